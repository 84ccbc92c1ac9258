"""Kernel-level parity on a real MI355X: every C-ABI entry point of libmft_hip.so against a float64
PyTorch-CPU statement of the same op (tolerances written per test).  `-m gpu` only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import meta_fine_tuning_amd  # noqa: F401
from meta_fine_tuning_amd import ops, synthetic

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_SHAPES = [
    # name, Cin, Cout, k, stride, pad, H
    ("trunk.0", 3, 64, 7, 2, 3, 84),
    ("trunk.4.C1", 64, 64, 3, 1, 1, 21),
    ("trunk.5.C1", 64, 128, 3, 2, 1, 21),
    ("trunk.5.C2", 128, 128, 3, 1, 1, 11),
    ("trunk.5.shortcut", 64, 128, 1, 2, 0, 21),
    ("trunk.6.C1", 128, 256, 3, 2, 1, 11),
    ("trunk.6.C2", 256, 256, 3, 1, 1, 6),
    ("trunk.6.shortcut", 128, 256, 1, 2, 0, 11),
    ("trunk.7.C1", 256, 512, 3, 2, 1, 6),
    ("trunk.7.C2", 512, 512, 3, 1, 1, 3),
    ("trunk.7.shortcut", 256, 512, 1, 2, 0, 6),
    ("trunk.7.C2@224", 512, 512, 3, 1, 1, 7),
    ("trunk.0@224", 3, 64, 7, 2, 3, 224),
]


@pytest.mark.parametrize("name,Cin,Cout,k,stride,pad,H", CONV_SHAPES)
def test_conv2d_forward(name, Cin, Cout, k, stride, pad, H):
    n = 5 if H < 200 else 2
    x = rnd((n, Cin, H, H), 1)
    w = rnd((Cout, Cin, k, k), 2, scale=(2.0 / (k * k * Cout)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    xg = ops.nchw_to_nhwc(x.to(DEV)) if Cin == 3 else nhwc(x).to(DEV)
    wpk = ops.pack_conv_weight(w.to(DEV))
    y = ops.conv2d(xg, wpk, Cout, k, k, stride, pad)
    got = nchw(y.cpu()).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * max(scale, 1.0), name


def test_conv2d_grouped_weights():
    """Per-episode weights: 3 groups x 5 images of trunk.7.C2 / C1 shapes; M tiles never straddle groups."""
    G, ipg = 3, 5
    for (Cin, Cout, k, stride, pad, H) in ((512, 512, 3, 1, 1, 3), (256, 512, 3, 2, 1, 6), (256, 512, 1, 2, 0, 6)):
        x = rnd((G * ipg, Cin, H, H), 3)
        w = rnd((G, Cout, Cin, k, k), 4, scale=0.05)
        ref = torch.cat([F.conv2d(x[g * ipg:(g + 1) * ipg].double(), w[g].double(), None, stride, pad) for g in range(G)])
        wpk = torch.stack([ops.pack_conv_weight(w[g].to(DEV)) for g in range(G)])
        y = ops.conv2d(nhwc(x).to(DEV), wpk, Cout, k, k, stride, pad, imgs_per_group=ipg)
        got = nchw(y.cpu()).double()
        assert float((got - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("M,K,N", [(13500, 133, 192), (13500, 192, 96), (450, 266, 48), (450, 458, 5), (13500, 96, 1),
                                   (100, 512, 128), (7, 32, 33)])
def test_gemm_padded(M, K, N):
    Kp = ops.round_up(K, 32)
    a = torch.zeros(M, Kp + 32)
    a[:, :K] = rnd((M, K), 5)
    a[:, K:] = 7.0                       # junk beyond K inside the padded row must be masked by zero weights
    w = rnd((N, K), 6, scale=K ** -0.5)
    b = rnd((N,), 7)
    ref = a[:, :K].double() @ w.double().t() + b.double()
    wpk = ops.pack_conv_weight(w.to(DEV))
    assert wpk.shape == (N, Kp)
    y = ops.gemm(a.to(DEV), Kp, wpk, N, bias=b.to(DEV))
    assert float((y.cpu().double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("M,K,N", [(105, 512, 128), (480, 266, 48), (480, 362, 48), (480, 458, 5), (1920, 458, 48), (7, 32, 33), (1, 16, 1), (33, 500, 70)])
def test_gemm_register_k_skinny(M, K, N):
    """mft_gemm_rk (the head's linear layers in a meta-training step: 16-row tiles, the whole K in registers over four waves) against
    float64 and against the tile kernel; junk beyond K in the operand rows is masked by the pack's zero columns; columns N.. of a
    caller-provided output are left alone; two runs are bit-identical."""
    Kp = ops.round_up(K, 32)
    a = torch.zeros(M, Kp + 32)
    a[:, :K] = rnd((M, K), 5)
    a[:, K:] = 7.0
    w = rnd((N, K), 6, scale=K ** -0.5)
    b = rnd((N,), 7)
    ref = a[:, :K].double() @ w.double().t() + b.double()
    wpk = ops.pack_conv_weight(w.to(DEV))
    ad, bd = a.to(DEV), b.to(DEV)
    y = ops.gemm_rk(ad, Kp, wpk, N, bias=bd)
    y2 = ops.gemm_rk(ad, Kp, wpk, N, bias=bd)
    yt = ops.gemm(ad, Kp, wpk, N, bias=bd)
    assert torch.equal(y, y2)
    tol = 2e-5 * max(float(ref.abs().max()), 1.0)
    assert float((y.cpu().double() - ref).abs().max()) <= tol
    assert float((y - yt).abs().max()) <= tol
    ld = ops.round_up(N, 32) + 32
    out = torch.full((M, ld), -3.0, device=DEV)
    ops.gemm_rk(ad, Kp, wpk, N, bias=None, out=out)
    assert float((out[:, :N].cpu().double() + b.double() - ref).abs().max()) <= tol and bool((out[:, N:] == -3.0).all())


def test_pack_roundtrip_and_dgrad():
    Cout, Cin, k, H, n = 512, 512, 3, 3, 5
    w = rnd((Cout, Cin, k, k), 8, scale=0.02)
    wpk = ops.pack_conv_weight(w.to(DEV))
    assert torch.equal(ops.unpack_conv_weight(wpk, (Cout, Cin, k, k)).cpu(), w)
    dy = rnd((n, Cout, H, H), 9)
    ref = torch.nn.grad.conv2d_input((n, Cin, H, H), w.double(), dy.double(), stride=1, padding=1)
    wt = ops.pack_dgrad_weight(wpk.view(1, Cout, -1), Cout, Cin, k, k, 1)[0]
    dx = ops.conv2d(nhwc(dy).to(DEV), wt, Cin, k, k, 1, 1)
    got = nchw(dx.cpu()).double()
    assert float((got - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)
    # direct dgrad from the forward pack (no transpose pass), shared and per-group weights
    dx2 = ops.conv2d_dgrad(nhwc(dy).to(DEV), wpk, Cin, k, k, 1)
    assert float((nchw(dx2.cpu()).double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("Cin,Cout,H,G,ipg", [(512, 512, 3, 3, 5), (512, 512, 7, 2, 5), (64, 64, 21, 1, 4), (128, 256, 6, 1, 5)])
def test_conv2d_dgrad_direct(Cin, Cout, H, G, ipg):
    k = 3
    w = rnd((G, Cout, Cin, k, k), 40, scale=0.05)
    dy = rnd((G * ipg, Cout, H, H), 41)
    ref = torch.cat([torch.nn.grad.conv2d_input((ipg, Cin, H, H), w[g].double(), dy[g * ipg:(g + 1) * ipg].double(),
                                                stride=1, padding=1) for g in range(G)])
    wpk = torch.stack([ops.pack_conv_weight(w[g].to(DEV)) for g in range(G)])
    dx = ops.conv2d_dgrad(nhwc(dy).to(DEV), wpk if G > 1 else wpk[0], Cin, k, k, 1, imgs_per_group=ipg if G > 1 else 0)
    assert float((nchw(dx.cpu()).double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H", [(64, 128, 3, 2, 1, 21), (64, 128, 1, 2, 0, 21), (256, 512, 3, 2, 1, 6),
                                                     (128, 256, 1, 2, 0, 11), (256, 512, 3, 2, 1, 14)])
def test_conv2d_dgrad_strided(Cin, Cout, k, stride, pad, H):
    n = 4
    OH = (H + 2 * pad - k) // stride + 1
    w = rnd((Cout, Cin, k, k), 42, scale=0.05)
    dy = rnd((n, Cout, OH, OH), 43)
    ref = torch.nn.grad.conv2d_input((n, Cin, H, H), w.double(), dy.double(), stride=stride, padding=pad)
    dx = ops.conv2d_dgrad(nhwc(dy).to(DEV), ops.pack_conv_weight(w.to(DEV)), Cin, k, k, pad, stride=stride, in_hw=(H, H))
    assert float((nchw(dx.cpu()).double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


def test_linear_dgrad_wgrad_odd_dims():
    """nn.Linear / 1x1-conv gradients of the GNN head: K padded to 32, N not a multiple of 64."""
    for (M, K, N) in ((14400, 133, 192), (14400, 192, 96), (480, 266, 48), (3000, 96, 4)):
        Kp = ops.round_up(K, 32)
        Np = ops.round_up(N, 32)
        x = torch.zeros(M, Kp); x[:, :K] = rnd((M, K), 44)
        dy = torch.zeros(M, Np); dy[:, :N] = rnd((M, N), 45)
        w = rnd((N, K), 46, scale=0.1)
        wp = torch.zeros(Np, Kp); wp[:N, :K] = w
        dx = ops.conv2d_dgrad(dy.to(DEV).view(M, 1, 1, Np), wp.to(DEV), Kp, 1, 1, 0)
        ref = dy[:, :N].double() @ w.double()
        assert float((dx.view(M, Kp)[:, :K].cpu().double() - ref).abs().max()) <= 3e-5 * max(float(ref.abs().max()), 1.0)
        dw = ops.conv2d_wgrad(x.to(DEV).view(M, 1, 1, Kp), dy.to(DEV).view(M, 1, 1, Np), Np, 1, 1, 1, 0)
        refw = dy[:, :N].double().t() @ x[:, :K].double()
        assert float((dw[0, :N, :K].cpu().double() - refw).abs().max()) <= 1e-4 * max(float(refw.abs().max()), 1.0)


def test_stem_and_large_m_wgrad():
    n, H = 21, 84
    x = rnd((n, 3, H, H), 47)
    dy = rnd((n, 64, 42, 42), 48)
    ref = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 7, 7), dy.double(), stride=2, padding=3)
    dw = ops.conv2d_wgrad(ops.nchw_to_nhwc(x.to(DEV)), nhwc(dy).to(DEV), 64, 7, 7, 2, 3)
    got = dw[0, :, :147].cpu().view(64, 7, 7, 3).permute(0, 3, 1, 2).double()
    assert float((got - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 1.0)
    x2, dy2 = rnd((n, 64, 21, 21), 49), rnd((n, 64, 21, 21), 50)          # 9261 rows -> 10 chunks
    ref2 = torch.nn.grad.conv2d_weight(x2.double(), (64, 64, 3, 3), dy2.double(), stride=1, padding=1)
    dw2 = ops.conv2d_wgrad(nhwc(x2).to(DEV), nhwc(dy2).to(DEV), 64, 3, 3, 1, 1)
    got2 = dw2[0].cpu().view(64, 3, 3, 64).permute(0, 3, 1, 2).double()
    assert float((got2 - ref2).abs().max()) <= 2e-4 * max(float(ref2.abs().max()), 1.0)


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H,G,ipg", [
    (512, 512, 3, 1, 1, 3, 3, 5), (256, 512, 3, 2, 1, 6, 3, 5), (256, 512, 1, 2, 0, 6, 2, 5),
    (512, 512, 3, 1, 1, 7, 1, 5), (64, 64, 3, 1, 1, 21, 1, 4), (256, 512, 3, 2, 1, 6, 1, 1)])
def test_conv2d_wgrad(Cin, Cout, k, stride, pad, H, G, ipg):
    OH = (H + 2 * pad - k) // stride + 1
    x = rnd((G * ipg, Cin, H, H), 10)
    dy = rnd((G * ipg, Cout, OH, OH), 11)
    ref = torch.stack([torch.nn.grad.conv2d_weight(x[g * ipg:(g + 1) * ipg].double(), (Cout, Cin, k, k),
                                                   dy[g * ipg:(g + 1) * ipg].double(), stride=stride, padding=pad)
                       for g in range(G)])
    dw = ops.conv2d_wgrad(nhwc(x).to(DEV), nhwc(dy).to(DEV), Cout, k, k, stride, pad, imgs_per_group=ipg)
    got = dw.cpu().view(G, Cout, k, k, Cin).permute(0, 1, 4, 2, 3).double()
    assert float((got - ref).abs().max()) <= 3e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("rows,C,G", [(45, 512, 4), (2205, 64, 3), (8820, 64, 1), (13500, 192, 1), (450, 48, 2),
                                      (100, 128, 1), (1, 512, 2)])
def test_bn_stats_apply_backward(rows, C, G):
    x = rnd((G * rows, C), 12) * 2.0 + 3.0            # non-zero mean: exercises the shifted-moment path
    gamma, beta = rnd((C,), 13).abs() + 0.5, rnd((C,), 14)
    res = rnd((G * rows, C), 15)
    dy = rnd((G * rows, C), 16)
    xg = x.to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, rstd = ops.bn_stats(xg, C, rows, G, rm if G == 1 else None, rv if G == 1 else None)
    xd = x.double().view(G, rows, C)
    mref = xd.mean(1)
    vref = xd.var(1, unbiased=False)
    assert float((mean.cpu().double() - mref).abs().max()) < 1e-5
    np.testing.assert_allclose(rstd.cpu().double().numpy(), (1.0 / torch.sqrt(vref + 1e-5)).numpy(), rtol=2e-5)
    if G == 1 and rows > 1:
        np.testing.assert_allclose(rm.cpu().numpy(), (0.1 * mref[0]).float().numpy(), atol=1e-6)
        np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * xd.var(1, unbiased=True)[0]).float().numpy(), rtol=2e-5)
    # apply (+ residual, relu)
    y = ops.bn_apply(xg, C, rows, G, mean, rstd, gamma.to(DEV), beta.to(DEV), act=ops.ACT_RELU, res=res.to(DEV))
    xhat = (xd - mref[:, None]) / torch.sqrt(vref[:, None] + 1e-5)
    yref = torch.relu(xhat * gamma.double() + beta.double() + res.double().view(G, rows, C))
    assert float((y.cpu().double().view(G, rows, C) - yref).abs().max()) < 2e-5
    y2 = ops.bn_apply(xg, C, rows, G, mean, rstd, gamma.to(DEV), beta.to(DEV), act=ops.ACT_LRELU)
    y2ref = F.leaky_relu(xhat * gamma.double() + beta.double(), 0.01)
    assert float((y2.cpu().double().view(G, rows, C) - y2ref).abs().max()) < 2e-5
    # backward through relu(bn(x)+res)
    if rows > 1:
        xa = x.double().view(G, rows, C).clone().requires_grad_(True)
        ga = gamma.double().clone().requires_grad_(True)
        ba = beta.double().clone().requires_grad_(True)
        mu = xa.mean(1, keepdim=True)
        va = xa.var(1, unbiased=False, keepdim=True)
        out = torch.relu((xa - mu) / torch.sqrt(va + 1e-5) * ga + ba + res.double().view(G, rows, C))
        gx, = torch.autograd.grad(out, xa, dy.double().view(G, rows, C), retain_graph=True)
        dx, dg, db = ops.bn_backward(xg, dy.to(DEV), C, rows, G, mean, rstd, gamma.to(DEV), relu_out=y)
        assert float((dx.cpu().double().view(G, rows, C) - gx).abs().max()) < 5e-5 * max(1.0, float(gx.abs().max()))
        for g in range(G):
            gg, gb = torch.autograd.grad(out[g], [ga, ba], dy.double().view(G, rows, C)[g], retain_graph=True)
            assert float((dg[g].cpu().double() - gg).abs().max()) < 1e-4 * max(1.0, float(gg.abs().max()))
            assert float((db[g].cpu().double() - gb).abs().max()) < 1e-4 * max(1.0, float(gb.abs().max()))


@pytest.mark.parametrize("rows,C,G", [(945, 512, 1), (3780, 256, 1), (945, 512, 4), (105, 128, 1), (480, 48, 2), (1, 64, 2), (4096, 64, 1)])
def test_bn_forward_small_one_launch(rows, C, G):
    """mft_bn_forward_small (statistics + running update + apply of a small train-mode BatchNorm in ONE launch; SimpleBlock's tail with
    the shortcut's own BatchNorm in the same launch) against float64 and against the launches it replaces (mft_bn_stats /
    mft_bn_stats_multi + mft_bn_apply): mean, rstd, running statistics (group 0 advances them), counter, output."""
    assert ops.bn_forward_small_ok(48, 480) and not ops.bn_forward_small_ok(48, 100000) and not ops.bn_forward_small_ok(6, 480)
    ld = ops.round_up(C, 32)
    x = torch.zeros(G * rows, ld)
    x[:, :C] = rnd((G * rows, C), 31) * 2.0 + 3.0
    r = rnd((G * rows, C), 32) * 0.7 - 1.0
    gamma, beta, rgam, rbet = rnd((C,), 33).abs() + 0.5, rnd((C,), 34), rnd((C,), 35).abs() + 0.5, rnd((C,), 36)
    xg, rg_ = x.to(DEV), r.to(DEV)

    def running():
        return torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
    ra, rb = running(), running()
    y, m, s, mr, sr = ops.bn_forward_small(xg, C, rows, G, gamma.to(DEV), beta.to(DEV), act=ops.ACT_RELU, running=ra, res=rg_,
                                           res_bn=(rgam.to(DEV), rbet.to(DEV), rb))
    xd, rd = x[:, :C].double().view(G, rows, C), r.double().view(G, rows, C)

    def stats(t):
        return t.mean(1), t.var(1, unbiased=False)
    (mx, vx), (mrr, vr) = stats(xd), stats(rd)
    assert float((m.cpu().double() - mx).abs().max()) < 1e-5 and float((mr.cpu().double() - mrr).abs().max()) < 1e-5
    np.testing.assert_allclose(s.cpu().double().numpy(), (1.0 / torch.sqrt(vx + 1e-5)).numpy(), rtol=2e-5)
    np.testing.assert_allclose(sr.cpu().double().numpy(), (1.0 / torch.sqrt(vr + 1e-5)).numpy(), rtol=2e-5)
    ref = torch.relu((xd - mx[:, None]) / torch.sqrt(vx[:, None] + 1e-5) * gamma.double() + beta.double() +
                     (rd - mrr[:, None]) / torch.sqrt(vr[:, None] + 1e-5) * rgam.double() + rbet.double())
    assert float((y[:, :C].cpu().double().view(G, rows, C) - ref).abs().max()) < 3e-5
    assert int(ra[2]) == int(rb[2]) == 1
    if rows > 1:
        np.testing.assert_allclose(ra[0].cpu().numpy(), (0.1 * mx[0]).float().numpy(), atol=1e-6)
        np.testing.assert_allclose(ra[1].cpu().numpy(), (0.9 + 0.1 * xd.var(1, unbiased=True)[0]).float().numpy(), rtol=2e-5)
        np.testing.assert_allclose(rb[1].cpu().numpy(), (0.9 + 0.1 * rd.var(1, unbiased=True)[0]).float().numpy(), rtol=2e-5)
    # against the launches it replaces
    m0, s0 = ops.bn_stats(xg, C, rows, G)
    m1, s1 = ops.bn_stats(rg_, C, rows, G)
    y0 = ops.bn_apply(xg, C, rows, G, m0, s0, gamma.to(DEV), beta.to(DEV), act=ops.ACT_RELU, res=rg_, res_bn=(m1, s1, rgam.to(DEV), rbet.to(DEV)))
    assert float((m - m0).abs().max()) < 1e-5 and float((s / s0 - 1).abs().max()) < 2e-5
    assert float((y[:, :C] - y0[:, :C]).abs().max()) < 3e-5
    # plain residual, no activation, no running statistics; leaky_relu without residual; run twice -> bit-identical
    y2, m2, s2 = ops.bn_forward_small(xg, C, rows, G, gamma.to(DEV), beta.to(DEV), act=ops.ACT_NONE, res=rg_)
    ref2 = (xd - mx[:, None]) / torch.sqrt(vx[:, None] + 1e-5) * gamma.double() + beta.double() + rd
    assert float((y2[:, :C].cpu().double().view(G, rows, C) - ref2).abs().max()) < 3e-5
    y3, _, _ = ops.bn_forward_small(xg, C, rows, G, gamma.to(DEV), beta.to(DEV), act=ops.ACT_LRELU)
    y4, _, _ = ops.bn_forward_small(xg, C, rows, G, gamma.to(DEV), beta.to(DEV), act=ops.ACT_LRELU)
    ref3 = F.leaky_relu((xd - mx[:, None]) / torch.sqrt(vx[:, None] + 1e-5) * gamma.double() + beta.double(), 0.01)
    assert float((y3[:, :C].cpu().double().view(G, rows, C) - ref3).abs().max()) < 3e-5 and torch.equal(y3[:, :C], y4[:, :C])
    # into a column window of a wider matrix at an odd offset (Gconv's output appended to the node features): 4-byte stores, same values
    if C <= 128:
        wide, off = torch.full((G * rows, 320), -7.0, device=DEV), 133
        ops.bn_forward_small(xg, C, rows, G, gamma.to(DEV), beta.to(DEV), act=ops.ACT_LRELU, out=wide, out_col=off)
        assert torch.equal(wide[:, off:off + C], y3[:, :C]) and bool((wide[:, :off] == -7.0).all()) and bool((wide[:, off + C:] == -7.0).all())


@pytest.mark.parametrize("rows,C,G", [(945, 512, 1), (3780, 256, 1), (945, 512, 2), (480, 48, 2)])
def test_bn_backward_small_matches_three_phase_form(rows, C, G):
    """The one-launch BatchNorm backward that mft_bn_backward_act(_multi) takes for small problems against the three-launch form on the
    same inputs (test hook 11000 + r: small up to r rows per group, 11000: never): dx, dgamma, dbeta to rounding (the sums are taken in another fixed order)."""
    from meta_fine_tuning_amd import _lib
    from meta_fine_tuning_amd import functional_bwd as FB
    x = rnd((G * rows, C), 41).to(DEV) * 1.5 + 0.3
    dy, ya = rnd((G * rows, C), 42).to(DEV), torch.relu(rnd((G * rows, C), 43)).to(DEV)
    ga = rnd((C,), 44).to(DEV) + 1.5
    m, s = ops.bn_stats(x, C, rows, G)
    assert _lib.lib().mft_debug_set_conv_tile(11000 + 4096) == 0          # (default: up to 512 rows per group)
    small = FB.bn_bwd(x, dy, C, G * rows, m, s, ga, y_act=ya, act=ops.ACT_RELU, groups=G)
    small2 = FB.bn_bwd(x, dy, C, G * rows, m, s, ga, y_act=ya, act=ops.ACT_RELU, groups=G)
    assert _lib.lib().mft_debug_set_conv_tile(11000) == 0
    big = FB.bn_bwd(x, dy, C, G * rows, m, s, ga, y_act=ya, act=ops.ACT_RELU, groups=G)
    _lib.lib().mft_debug_reset()
    for a, b, c in zip(small, big, small2):
        assert torch.equal(a, c)
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), float((a - b).abs().max())


def test_bn_apply_residual_bn():
    rows, C, G = 45, 512, 3
    x, r = rnd((G * rows, C), 17), rnd((G * rows, C), 18)
    g1, b1, g2, b2 = (rnd((G, C), s) for s in (19, 20, 21, 22))     # per-group affine (per-episode last block)
    xg, rg = x.to(DEV), r.to(DEV)
    m1, s1 = ops.bn_stats(xg, C, rows, G)
    m2, s2 = ops.bn_stats(rg, C, rows, G)
    y = ops.bn_apply(xg, C, rows, G, m1, s1, g1.to(DEV), b1.to(DEV), act=ops.ACT_RELU, res=rg,
                     res_bn=(m2, s2, g2.to(DEV), b2.to(DEV)), gb_group_stride=C)
    xd, rd = x.double().view(G, rows, C), r.double().view(G, rows, C)

    def bn(t, g, b):
        return (t - t.mean(1, keepdim=True)) / torch.sqrt(t.var(1, unbiased=False, keepdim=True) + 1e-5) * g[:, None].double() + b[:, None].double()
    ref = torch.relu(bn(xd, g1, b1) + bn(rd, g2, b2))
    assert float((y.cpu().double().view(G, rows, C) - ref).abs().max()) < 3e-5


def test_stem_tail_and_pools():
    n, H, C = 10, 42, 64
    x = rnd((n, C, H, H), 23)
    gamma, beta = rnd((C,), 24).abs() + 0.5, rnd((C,), 25)
    xg = nhwc(x).to(DEV)
    ipg = 5
    mean, rstd = ops.bn_stats(xg.view(-1, C), C, ipg * H * H, n // ipg)
    y = ops.bn_relu_maxpool(xg, mean, rstd, gamma.to(DEV), beta.to(DEV), imgs_per_group=ipg)
    refs = []
    for g in range(n // ipg):
        xx = x[g * ipg:(g + 1) * ipg].double()
        bn = F.batch_norm(xx, None, None, gamma.double(), beta.double(), True, 0.0, 1e-5)
        refs.append(F.max_pool2d(torch.relu(bn), 3, 2, 1))
    ref = torch.cat(refs)
    assert float((nchw(y.cpu()).double() - ref).abs().max()) < 2e-5
    # global avgpool + its backward fused with relu backward
    o = torch.relu(rnd((n, 512, 3, 3), 26))
    og = nhwc(o).to(DEV)
    f = ops.global_avgpool(og)
    assert float((f.cpu().double() - o.double().mean((2, 3))).abs().max()) < 1e-6
    df = rnd((n, 512), 27)
    d = ops.avgpool_relu_backward(df.to(DEV), og)
    ref = (o > 0).double() * df.double()[:, :, None, None] / 9.0
    assert float((nchw(d.cpu()).double() - ref).abs().max()) < 1e-7


def test_cross_entropy_softmax():
    G, rpg, C = 4, 5, 512
    x = rnd((G * rpg, C), 28) * 3
    y = torch.from_numpy(np.random.RandomState(29).randint(0, 5, size=(G * rpg,)))
    loss, d = ops.cross_entropy(x.to(DEV), y.to(torch.int32).to(DEV), rpg, G)
    for g in range(G):
        xa = x[g * rpg:(g + 1) * rpg].double().requires_grad_(True)
        l = F.cross_entropy(xa, y[g * rpg:(g + 1) * rpg])
        gx, = torch.autograd.grad(l, xa)
        assert abs(float(loss[g].cpu()) - float(l)) < 1e-5
        assert float((d[g * rpg:(g + 1) * rpg].cpu().double() - gx).abs().max()) < 1e-6
    s = ops.softmax_rows(x[:75, :5].contiguous().to(DEV))
    assert float((s.cpu().double() - F.softmax(x[:75, :5].double(), 1)).abs().max()) < 1e-6


def test_adam_sgd_maml():
    n = 3_673_088 + 3
    p, g = rnd((n,), 30, 0.05), rnd((n,), 31, 1e-3)
    pd, md, vd = p.double().clone(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    pg, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for t in range(1, 4):
        gt = g * t
        ops.adam_step(pg, gt.to(DEV), m, v, t, lr=0.01)
        gd = gt.double()
        md = 0.9 * md + 0.1 * gd
        vd = 0.999 * vd + 0.001 * gd * gd
        pd = pd - (0.01 / (1 - 0.9 ** t)) * md / (vd.sqrt() / (1 - 0.999 ** t) ** 0.5 + 1e-8)
    assert float((pg.cpu().double() - pd).abs().max()) < 1e-6
    # SGD with momentum/dampening/wd
    p2, buf = p[:2565].clone().to(DEV), torch.zeros(2565, device=DEV)
    pr, br = p[:2565].double().clone(), None
    for t in range(3):
        gt = g[:2565] * (t + 1)
        ops.sgd_step(p2, gt.to(DEV), buf, first_step=(t == 0))
        ge = gt.double() + 0.001 * pr
        br = ge.clone() if br is None else 0.9 * br + 0.1 * ge
        pr = pr - 0.01 * br
    assert float((p2.cpu().double() - pr).abs().max()) < 1e-7
    a, b, c = rnd((1000,), 32), rnd((1000,), 33), rnd((1000,), 34)
    ag = a.to(DEV)
    ops.maml_delta(ag, b.to(DEV), c.to(DEV))
    assert torch.equal(ag.cpu(), a - (c - b))


@pytest.mark.parametrize("B,N,Fd", [(15, 30, 133), (2, 105, 181), (2, 130, 229), (130, 30, 133), (128, 130, 229), (129, 7, 181)])
def test_gnn_glue(B, N, Fd):
    # B >= 128 graphs: mft_graph_aggregate's one-workgroup-per-graph form with x[b] staged in LDS
    ld = 256
    x = torch.zeros(B * N, ld)
    x[:, :Fd] = rnd((B * N, Fd), 35)
    Kp = ops.round_up(Fd, 32)
    d = ops.pair_absdiff(x.to(DEV), N, Fd, Kp)
    xv = x[:, :Fd].view(B, N, Fd)
    ref = (xv.unsqueeze(2) - xv.unsqueeze(1)).abs().reshape(B * N * N, Fd)
    assert torch.equal(d.cpu()[:, :Fd], ref) and float(d.cpu()[:, Fd:].abs().max()) == 0.0
    s = torch.zeros(B * N * N, 4)
    s[:, 0] = rnd((B * N * N,), 36) * 2
    A = ops.masked_softmax(s.to(DEV), N)
    sref = s[:, 0].double().view(B, N, N) - torch.eye(N, dtype=torch.float64) * 1e8
    Aref = F.softmax(sref, 2)
    assert float((A.cpu().double() - Aref).abs().max()) < 1e-6
    y = ops.graph_aggregate(A, x.to(DEV), Fd, ops.round_up(2 * Fd, 32))
    yref = torch.cat([xv.double(), torch.bmm(A.cpu().double(), xv.double())], 2).view(B * N, 2 * Fd)
    assert float((y.cpu().double()[:, :2 * Fd] - yref).abs().max()) < 1e-5
    assert float(y.cpu()[:, 2 * Fd:].abs().max()) == 0.0
    z = torch.zeros(B * N, ld, device=DEV)
    ops.copy_cols(y, z, 7, 48, act=ops.ACT_LRELU)
    assert float((z.cpu()[:, 7:55].double() - F.leaky_relu(y.cpu()[:, :48].double(), 0.01)).abs().max()) < 1e-7


@pytest.mark.parametrize("name,Cin,Cout,k,stride,pad,H", [c for c in CONV_SHAPES if c[1] % 32 == 0 and c[2] % 64 == 0])
def test_conv2d_bf16x3_is_fp32_accurate(name, Cin, Cout, k, stride, pad, H):
    """6-term bf16x3 convolution (bf16 MFMA) against float64: same tolerance as the fp32-MFMA kernel, and an error no
    larger than 2x that kernel's own error (the split is exact, dropped terms are <= 2^-24 relative)."""
    n = 7 if H < 200 else 2
    x = rnd((n, Cin, H, H), 11)
    w = rnd((Cout, Cin, k, k), 12, scale=(2.0 / (k * k * Cout)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    xg = nhwc(x).to(DEV)
    wpk = ops.pack_conv_weight(w.to(DEV))
    from meta_fine_tuning_amd import _lib
    OH = (H + 2 * pad - k) // stride + 1
    y32g = torch.empty((n, OH, OH, Cout), device=DEV)          # the fp32-MFMA kernel walking K in one piece (ops.conv2d slices K on small batches)
    assert _lib.lib().mft_conv2d_nhwc(ops._p(xg), Cin, ops._p(wpk), None, ops._p(y32g), Cout, n, H, H, Cin, Cout, k, k, stride, pad, 0, 0,
                                      ops._stream()) == 0
    y32 = nchw(y32g.cpu()).double()
    scale = max(float(ref.abs().max()), 1.0)
    e32 = float((y32 - ref).abs().max())
    w3 = ops.split_weight_x3(wpk)
    # default (shared-tap form on 3x3 / stride-1 layers), per-tap implicit-GEMM form, 64x64 tiles; in an MFT_EXPERIMENTS build also the
    # measured-slower variants: LDS-patch form, 512-thread ping-pong form, the two pinned fragment-read schedules, XOR-swizzled LDS layout
    modes = (10, 90, 3) + ((12, 71, 81, 82, 42) if _lib.lib().mft_has_experiments() else ())
    if not _lib.lib().mft_has_experiments():
        assert _lib.lib().mft_debug_set_x3_tile(71) == _lib.MFT_EINVAL and _lib.lib().mft_debug_set_x3_tile(42) == _lib.MFT_EINVAL
    for patch_mode in modes:
        _lib.lib().mft_debug_reset()
        assert _lib.lib().mft_debug_set_x3_tile(patch_mode) == 0
        y3 = nchw(ops.conv2d_x3(xg, w3, Cout, k, k, stride, pad).cpu()).double()
        e3 = float((y3 - ref).abs().max())
        assert e3 <= 2e-5 * scale, (name, patch_mode, e3)
        assert e3 <= 2.0 * e32 + 1e-7 * scale, (name, patch_mode, e3, e32)
    _lib.lib().mft_debug_reset()


@pytest.mark.parametrize("xscale", [1.0, 1e3, 1e-3])
@pytest.mark.parametrize("name,Cin,Cout,k,stride,pad,H", [c for c in CONV_SHAPES if c[1] % 32 == 0 and c[2] % 64 == 0 and c[0].startswith(("trunk.4", "trunk.5", "trunk.6"))])
def test_conv2d_f16x2_is_fp32_accurate(name, Cin, Cout, k, stride, pad, H, xscale):
    """3-term f16x2 convolution (two fp16 pieces per operand, lo piece pre-scaled by 2^11, leading / cross products in separate fp32
    accumulators; csrc/conv_x3.hip NP = 2) against float64 on the frozen-trunk shapes: the tolerance of the fp32-MFMA kernel, and an
    error no larger than 2x that kernel's own (the same two gates as the bf16x3 form).  ReLU-shaped (non-negative) activations as in
    the trunk, at three magnitudes: the scaled lo piece keeps the RELATIVE accuracy wherever the hi piece is a normal fp16 number."""
    from meta_fine_tuning_amd import _lib
    n = 7
    x = torch.relu(rnd((n, Cin, H, H), 11) + 0.3) * xscale
    w = rnd((Cout, Cin, k, k), 12, scale=(2.0 / (k * k * Cout)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    xg = nhwc(x).to(DEV)
    wpk = ops.pack_conv_weight(w.to(DEV))
    OH = (H + 2 * pad - k) // stride + 1
    y32g = torch.empty((n, OH, OH, Cout), device=DEV)
    assert _lib.lib().mft_conv2d_nhwc(ops._p(xg), Cin, ops._p(wpk), None, ops._p(y32g), Cout, n, H, H, Cin, Cout, k, k, stride, pad, 0, 0,
                                      ops._stream()) == 0
    e32 = float((nchw(y32g.cpu()).double() - ref).abs().max())
    scale = max(float(ref.abs().max()), 1e-30)
    w2 = ops.split_weight_h2(wpk)
    assert w2.shape == (2,) + tuple(wpk.shape)
    y3 = nchw(ops.conv2d_x3(xg, ops.split_weight_x3(wpk), Cout, k, k, stride, pad).cpu()).double()
    for mode in (10, 90):                                     # default (shared-tap kernel on 3x3 / stride 1), per-tap kernel
        _lib.lib().mft_debug_reset()
        assert _lib.lib().mft_debug_set_x3_tile(mode) == 0
        y2 = nchw(ops.conv2d_x3(xg, w2, Cout, k, k, stride, pad).cpu()).double()
        e2 = float((y2 - ref).abs().max())
        assert e2 <= 2e-5 * scale, (name, mode, e2, scale)
        assert e2 <= 2.0 * e32 + 1e-7 * scale, (name, mode, e2, e32)
        assert float((y2 - y3).abs().max()) <= 4e-6 * scale      # and it agrees with the bf16x3 form
    _lib.lib().mft_debug_reset()
    # the 64x64-tile and other alternative forms exist for bf16x3 only: refused, never silently another kernel
    assert _lib.lib().mft_debug_set_x3_tile(3) == 0
    out = torch.empty((n, OH, OH, Cout), device=DEV)
    assert _lib.lib().mft_conv2d_nhwc_h2(ops._p(xg), Cin, ops._p(w2), w2.shape[1] * w2.shape[2], ops._p(out), Cout, n, H, H, Cin, Cout, k, k,
                                         stride, pad, ops._stream()) == _lib.MFT_EINVAL
    _lib.lib().mft_debug_reset()


def test_f16x2_split_and_range_guard():
    """mft_split_f16x2 against the same arithmetic in torch (hi = fp16(w), lo = fp16((w - hi) * 2^11): |w - hi - lo / 2^11| <= 2^-22 |w|
    wherever hi is a normal fp16 number), and functional.f16x2_safe: reference-style weights pass, a BatchNorm gamma or a weight that
    could leave fp16's range keeps the bf16x3 kernels."""
    from meta_fine_tuning_amd import functional as Fn
    w = torch.cat([rnd((4096,), 5) * 0.05, torch.tensor([0.0, 1.0, -1.0, 65000.0, 6.2e-5, 1e-7, -3e-6, 1.00048828125])])
    pl = ops.split_weight_h2(w.view(1, -1).to(DEV).contiguous()).cpu()
    hi, lo = pl[0, 0].view(torch.float16), pl[1, 0].view(torch.float16)
    hi_ref = w.to(torch.float16)
    lo_ref = ((w - hi_ref.float()) * 2048.0).to(torch.float16)
    assert torch.equal(hi, hi_ref) and torch.equal(lo, lo_ref)
    rec = hi.double() + lo.double() / 2048.0
    normal = w.abs() >= 6.2e-5
    assert float(((rec - w.double()).abs() / w.double().abs().clamp_min(1e-30))[normal].max()) <= 2.0 ** -22
    assert float((rec - w.double()).abs()[~normal].max()) <= 2.0 ** -35
    sd = synthetic.resnet10_state_dict(seed=3)
    assert Fn.f16x2_safe(sd)
    W = Fn.ResNet10Weights(sd, DEV, x3=True)
    assert W.f16x2 == Fn.TRUNK_F16X2 and W.conv3["trunk.4.C1"].shape[0] == (2 if Fn.TRUNK_F16X2 else 3)
    for key, val in (("trunk.5.BN1.weight", 40.0), ("trunk.4.BN2.bias", 4e4), ("trunk.6.C2.weight", 7e4), ("trunk.1.weight", float("nan")),
                     ("trunk.6.BN1.weight", None), ("trunk.5.C1.weight", None)):
        bad = dict(sd)
        bad[key] = sd[key].clone()
        if val is None:
            bad[key] *= 1e-5                                          # a whole tensor far below fp16's normal range
        else:
            bad[key].view(-1)[0] = val
        assert not Fn.f16x2_safe(bad), key
        assert Fn.ResNet10Weights(bad, DEV, x3=True).conv3["trunk.4.C1"].shape[0] == 3
    # the proof is made for the LARGEST BatchNorm group the caller runs: gamma = 14 passes for 2^20 rows
    # (14 * 1024 + |beta| < 3e4), not for the 4.08 M rows of a 50-shot 224 x 224 final pass (14 * 2020 = 28,280 + ...: the stem's)
    big = dict(sd)
    big["trunk.1.weight"] = sd["trunk.1.weight"].clone()
    big["trunk.1.weight"].view(-1)[0] = 14.5
    assert Fn.f16x2_safe(big) and not Fn.f16x2_safe(big, max_rows=325 * 112 * 112)
    # ... and the fp16 planes are handed out only inside it, and never behind eval-mode statistics
    if W.f16x2:
        Wr = Fn.ResNet10Weights(sd, DEV, x3=True, max_rows=Fn.stem_rows_bound(100, 84))
        assert Wr.f16x2 and all(Wr.planes(p, 100, h) is Wr.conv3 for p, h in (("trunk.4", 21), ("trunk.5", 21), ("trunk.6", 11)))
        assert Wr.planes("trunk.4", 111, 21) == {} and Wr.planes("trunk.5", 5, 21, running={"x": 1}) == {}
        assert Wr.planes("trunk.5", 5, 21, fixed={"x": 1}) == {} and Wr.planes("trunk.5", 5, 21) is Wr.conv3
        Wb = Fn.ResNet10Weights(sd, DEV, x3=True, f16x2=False)
        assert Wb.planes("trunk.5", 5, 21, running={"x": 1}) is Wb.conv3                                            # bf16x3: no range condition


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(7, 5, 5, 32, 64), (3, 13, 13, 64, 128), (1, 21, 21, 64, 64), (9, 7, 9, 96, 64), (2, 9, 4, 32, 64),
                                            (130, 1, 1, 32, 64), (5, 3, 40, 64, 64), (4, 130, 2, 32, 64)])
def test_conv2d_bf16x3_shared_tap_geometries(n, H, W, Cin, Cout):
    """conv_x3_s1_kernel (3x3, stride 1, pad 1: one staged image per (kh, channel slice) serves the three kw taps; zero rows at the
    image-row boundaries are the horizontal padding) on geometries that stress its row bookkeeping -- rows shorter and longer than a
    128-pixel tile, non-square maps, many images per tile, a ragged last tile, 1x1 maps -- against float64 and against the per-tap
    kernel (same products, other summation order), with and without the BatchNorm statistics epilogue."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    x = rnd((n, Cin, H, W), 61)
    w = rnd((Cout, Cin, 3, 3), 62, scale=(2.0 / (9 * Cout)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    xg = nhwc(x).to(DEV)
    w3 = ops.split_weight_x3(ops.pack_conv_weight(w.to(DEV)))
    scale = max(float(ref.abs().max()), 1.0)
    outs = {}
    for knob in (91, 90):
        lib.mft_debug_reset()
        lib.mft_debug_set_x3_tile(knob)
        outs[knob] = nchw(ops.conv2d_x3(xg, w3, Cout, 3, 3, 1, 1).cpu()).double()
        assert float((outs[knob] - ref).abs().max()) <= 2e-5 * scale, knob
    lib.mft_debug_reset()
    assert float((outs[91] - outs[90]).abs().max()) <= 4e-6 * scale
    if Cin > 32 and H * W > 1:                      # with one channel slice both kernels walk K in the same order
        assert not torch.equal(outs[91], outs[90])  # the shared-tap kernel really ran
    # statistics epilogue: groups of ipg images with >= 128 rows each
    ipg = max(1, -(-128 // (H * W)))
    if n % ipg == 0:
        groups = n // ipg
        out = torch.empty(n, H, W, Cout, device=DEV)
        nws = int(lib.mft_conv2d_x3_stats_ws_floats(n, H, W, Cout, 3, 3, 1, 1))
        ws = torch.empty(max(nws, 1), device=DEV)
        mean, rstd = torch.empty(groups, Cout, device=DEV), torch.empty(groups, Cout, device=DEV)
        r = ops.conv2d_x3_bnstats(xg, w3, Cout, 3, 3, 1, 1, ipg, out, ws, mean, rstd)
        if r is not None:
            yr = ref.permute(0, 2, 3, 1).reshape(groups, -1, Cout)
            assert float((mean.cpu().double() - yr.mean(1)).abs().max()) <= 1e-5 * scale
            assert float((rstd.cpu().double() - 1.0 / (yr.var(1, unbiased=False) + 1e-5).sqrt()).abs().max()) <= 2e-4 * float(rstd.abs().max())
            assert float((nchw(out.cpu()).double() - ref).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("ipg", [5, 4, 1])
@pytest.mark.parametrize("name,Cin,Cout,k,stride,pad,H", [("trunk.7.C1", 256, 512, 3, 2, 1, 6), ("trunk.7.C2", 512, 512, 3, 1, 1, 3),
                                                          ("trunk.7.shortcut", 256, 512, 1, 2, 0, 6)])
def test_skinny_per_episode_conv_and_dgrad(name, Cin, Cout, k, stride, pad, H, ipg):
    """Weight-streaming skinny kernels (csrc/skinny.hip; per-episode weights, <= 48 pixels per episode) against float64,
    and bit-for-bit routing check against the generic tiles (same tolerance)."""
    from meta_fine_tuning_amd import _lib
    G = 3
    n = G * ipg
    x = rnd((n, Cin, H, H), 21)
    w = rnd((G, Cout, Cin, k, k), 22, scale=(2.0 / (k * k * Cout)) ** 0.5)
    xg = nhwc(x).to(DEV)
    wpk = torch.stack([ops.pack_conv_weight(w[g].to(DEV)) for g in range(G)])
    ref = torch.cat([F.conv2d(x[g * ipg:(g + 1) * ipg].double(), w[g].double(), None, stride, pad) for g in range(G)])
    outs = []
    for mode in (3000, 3001, 8000):                   # generic tiles, skinny bf16x3 form (default), skinny fp32-MFMA form
        _lib.lib().mft_debug_set_conv_tile(mode)
        outs.append(nchw(ops.conv2d(xg, wpk, Cout, k, k, stride, pad, imgs_per_group=ipg).cpu()).double())
    _lib.lib().mft_debug_set_conv_tile(3001)
    _lib.lib().mft_debug_set_conv_tile(8001)
    # full-line weight loads + DPP re-deal (default) against the MFMA-fragment-order loads: the same registers end up with the same
    # values, so not a bit may change
    _lib.lib().mft_debug_set_conv_tile(9700)
    frag = nchw(ops.conv2d(xg, wpk, Cout, k, k, stride, pad, imgs_per_group=ipg).cpu()).double()
    _lib.lib().mft_debug_set_conv_tile(9701)
    assert torch.equal(frag, outs[1]), name
    scale = max(float(ref.abs().max()), 1.0)
    e32 = float((outs[2] - ref).abs().max())
    for o in outs:
        assert float((o - ref).abs().max()) <= 2e-5 * scale, name
    # the six-term bf16 product is as accurate as the fp32 matrix instruction it replaces
    assert float((outs[1] - ref).abs().max()) <= 2.0 * e32 + 1e-7 * scale, (name, e32)
    if stride == 1:
        dy = rnd((n, Cout, H, H), 23)
        dyg = nhwc(dy).to(DEV)
        xr = x.double().requires_grad_(True)
        refs = []
        for g in range(G):
            xi = x[g * ipg:(g + 1) * ipg].double().requires_grad_(True)
            F.conv2d(xi, w[g].double(), None, 1, pad).backward(dy[g * ipg:(g + 1) * ipg].double())
            refs.append(xi.grad)
        refd = torch.cat(refs)
        errs = {}
        for mode in (3000, 3001, 8000, 6002):         # generic tiles, skinny x3, skinny fp32, skinny fp32 with two LDS slices
            _lib.lib().mft_debug_set_conv_tile(mode)
            dx = nchw(ops.conv2d_dgrad(dyg, wpk, Cin, k, k, pad, imgs_per_group=ipg).cpu()).double()
            errs[mode] = float((dx - refd).abs().max())
            assert errs[mode] <= 2e-5 * max(float(refd.abs().max()), 1.0), (name, mode)
        assert errs[3001] <= 2.0 * errs[8000] + 1e-7 * max(float(refd.abs().max()), 1.0), (name, errs)
        _lib.lib().mft_debug_set_conv_tile(6001)
        _lib.lib().mft_debug_set_conv_tile(8001)
        _lib.lib().mft_debug_set_conv_tile(3001)


@pytest.mark.parametrize("ipg", [5, 3, 1])
def test_fused_small_group_kernels_match_unfused(ipg):
    """mft_bn_small_forward / mft_bn_backward2 / mft_ce_pool_backward (one launch each) against the launch sequences they
    replace (bn_stats + bn_apply + global_avgpool; two bn_backward; cross_entropy + avgpool_relu_backward)."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    G, C, hw = 4, 512, 9
    rows = ipg * hw
    n = G * ipg
    c2 = rnd((n * hw, C), 31).to(DEV)
    sc = rnd((n * hw, C), 32).to(DEV) * 0.5 + 0.3
    g2, b2 = (rnd((G, C), 33) * 0.2 + 1).to(DEV), (rnd((G, C), 34) * 0.1).to(DEV)
    gs, bs = (rnd((G, C), 35) * 0.2 + 1).to(DEV), (rnd((G, C), 36) * 0.1).to(DEV)
    # unfused
    m2, s2 = ops.bn_stats(c2, C, rows, G)
    ms, ss = ops.bn_stats(sc, C, rows, G)
    ref = ops.bn_apply(c2, C, rows, G, m2, s2, g2, b2, act=ops.ACT_RELU, res=sc, res_bn=(ms, ss, gs, bs), gb_group_stride=C)
    ref_feat = ops.global_avgpool(ref.view(n, 3, 3, C))
    # fused
    out = torch.empty_like(c2)
    feat = torch.empty((n, C), device=DEV)
    fm2, fs2, fms, fss = (torch.empty((G, C), device=DEV) for _ in range(4))
    rc = lib.mft_bn_small_forward(ops._p(c2), C, ops._p(sc), C, None, 0, ops._p(out), C, C, rows, G, ops._p(g2), ops._p(b2),
                                  ops._p(gs), ops._p(bs), C, ops._p(fm2), ops._p(fs2), ops._p(fms), ops._p(fss), ops.ACT_RELU,
                                  0.0, 1e-5, ops._p(feat), hw, ops._stream())
    assert rc == 0
    assert float((out - ref).abs().max()) < 2e-5 and float((feat - ref_feat).abs().max()) < 1e-5
    assert float((fm2 - m2).abs().max()) < 1e-6 and float(((fs2 - s2) / s2).abs().max()) < 1e-5
    assert float((fms - ms).abs().max()) < 1e-6 and float(((fss - ss) / ss).abs().max()) < 1e-5
    # CE + pool backward
    labels = torch.from_numpy(np.random.RandomState(37).randint(0, 5, size=n).astype(np.int32)).to(DEV)
    loss_ref, dl = ops.cross_entropy(ref_feat, labels, ipg, G)
    d_ref = ops.avgpool_relu_backward(dl, ref.view(n, 3, 3, C))
    d_out = torch.empty_like(ref)
    loss = torch.empty((G,), device=DEV)
    assert lib.mft_ce_pool_backward(ops._p(ref_feat), ops._p(labels), ipg, G, C, hw, ops._p(ref), ops._p(d_out), ops._p(loss),
                                    ops._stream()) == 0
    assert float((d_out.view(-1) - d_ref.reshape(-1)).abs().max()) < 1e-7 and float((loss - loss_ref).abs().max()) < 1e-5
    # dual BN backward
    dxa_r, dga_r, dba_r = ops.bn_backward(c2, d_out, C, rows, G, m2, s2, g2, gb_group_stride=C)
    dxb_r, dgb_r, dbb_r = ops.bn_backward(sc, d_out, C, rows, G, ms, ss, gs, gb_group_stride=C)
    dxa, dxb = torch.empty_like(c2), torch.empty_like(sc)
    dga, dba, dgb, dbb = (torch.empty((G, C), device=DEV) for _ in range(4))
    assert lib.mft_bn_backward2(ops._p(c2), ops._p(sc), C, ops._p(d_out), C, ops._p(dxa), ops._p(dxb), C, C, rows, G, ops._p(m2),
                                ops._p(s2), ops._p(g2), ops._p(ms), ops._p(ss), ops._p(gs), C, ops._p(dga), ops._p(dba),
                                ops._p(dgb), ops._p(dbb), ops._stream()) == 0
    for a, b in ((dxa, dxa_r), (dxb, dxb_r), (dga, dga_r), (dba, dba_r), (dgb, dgb_r), (dbb, dbb_r)):
        assert torch.equal(a, b)
    # CE + pool backward + dual BN backward in one launch: same arithmetic in the same order
    dxa2, dxb2 = torch.empty_like(c2), torch.empty_like(sc)
    dga2, dba2, dgb2, dbb2 = (torch.empty((G, C), device=DEV) for _ in range(4))
    loss2 = torch.empty((G,), device=DEV)
    assert lib.mft_ce_pool_bn_backward2(ops._p(ref_feat), ops._p(labels), ipg, G, C, hw, ops._p(ref), ops._p(c2), ops._p(sc),
                                        ops._p(dxa2), ops._p(dxb2), ops._p(m2), ops._p(s2), ops._p(g2), ops._p(ms), ops._p(ss),
                                        ops._p(gs), C, ops._p(dga2), ops._p(dba2), ops._p(dgb2), ops._p(dbb2), ops._p(loss2),
                                        ops._stream()) == 0
    for a, b in ((dxa2, dxa), (dxb2, dxb), (dga2, dga), (dba2, dba), (dgb2, dgb), (dbb2, dbb), (loss2, loss)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ipg", [5, 4, 1])
def test_fused_block_entry_and_dgrad_bn_backward(ipg, monkeypatch):
    """mft_block_entry_small_forward (C1 + BatchNorm + ReLU + shortcut conv in one launch) and
    mft_conv2d_dgrad_bn_backward_small (C2 data gradient + BatchNorm/ReLU backward) against the launch sequences they replace
    (the weight-streaming bf16x3 kernels of large episode batches: ops.SMALL_GROUPS = 0 keeps the three test episodes on them)."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    monkeypatch.setattr(ops, "SMALL_GROUPS", 0)
    G, Cin, C, H = 3, 256, 512, 6
    n = G * ipg
    rows = ipg * 9
    x = nhwc(rnd((n, Cin, H, H), 41)).to(DEV)
    w1 = torch.stack([ops.pack_conv_weight(rnd((C, Cin, 3, 3), 42 + g, scale=0.03).to(DEV)) for g in range(G)])
    wsc = torch.stack([ops.pack_conv_weight(rnd((C, Cin, 1, 1), 52 + g, scale=0.08).to(DEV)) for g in range(G)])
    g1, b1 = (rnd((G, C), 61) * 0.2 + 1).to(DEV), (rnd((G, C), 62) * 0.1).to(DEV)
    # unfused reference sequence
    c1_r = ops.conv2d(x, w1, C, 3, 3, 2, 1, imgs_per_group=ipg)
    sc_r = ops.conv2d(x, wsc, C, 1, 1, 2, 0, imgs_per_group=ipg)
    m_r, s_r = ops.bn_stats(c1_r.view(-1, C), C, rows, G)
    r1_r = ops.bn_apply(c1_r.view(-1, C), C, rows, G, m_r, s_r, g1, b1, act=ops.ACT_RELU, gb_group_stride=C)
    c1, sc, r1 = torch.empty_like(c1_r), torch.empty_like(sc_r), torch.empty_like(r1_r)
    m1, s1 = torch.empty((G, C), device=DEV), torch.empty((G, C), device=DEV)
    rc = lib.mft_block_entry_small_forward(ops._p(x), Cin, ops._p(w1), C * 9 * Cin, ops._p(wsc), C * Cin, ops._p(c1), ops._p(r1),
                                           ops._p(sc), n, H, H, Cin, C, 2, ipg, ops._p(g1), ops._p(b1), C, ops._p(m1), ops._p(s1),
                                           1e-5, ops._stream())
    assert rc == 0
    assert torch.equal(c1, c1_r) and torch.equal(sc, sc_r)            # same bf16x3 arithmetic in the same order
    assert float((m1 - m_r).abs().max()) < 1e-6 and float(((s1 - s_r) / s_r).abs().max()) < 1e-5
    assert float((r1 - r1_r).abs().max()) < 2e-5
    # exit half: C2 + BN2 + BN(shortcut) + add + ReLU + average pool
    w2 = torch.stack([ops.pack_conv_weight(rnd((C, C, 3, 3), 72 + g, scale=0.02).to(DEV)) for g in range(G)])
    g2, b2 = (rnd((G, C), 63) * 0.2 + 1).to(DEV), (rnd((G, C), 64) * 0.1).to(DEV)
    gs, bs = (rnd((G, C), 65) * 0.2 + 1).to(DEV), (rnd((G, C), 66) * 0.1).to(DEV)
    r1v = r1_r.view(n, 3, 3, C)
    c2_r = ops.conv2d(r1v, w2, C, 3, 3, 1, 1, imgs_per_group=ipg)
    out_r, feat_r = torch.empty((n * 9, C), device=DEV), torch.empty((n, C), device=DEV)
    st_r = [torch.empty((G, C), device=DEV) for _ in range(4)]
    assert lib.mft_bn_small_forward(ops._p(c2_r), C, ops._p(sc_r), C, None, 0, ops._p(out_r), C, C, rows, G, ops._p(g2), ops._p(b2),
                                    ops._p(gs), ops._p(bs), C, ops._p(st_r[0]), ops._p(st_r[1]), ops._p(st_r[2]), ops._p(st_r[3]),
                                    ops.ACT_RELU, 0.0, 1e-5, ops._p(feat_r), 9, ops._stream()) == 0
    c2, out, feat = torch.empty_like(c2_r), torch.empty_like(out_r), torch.empty_like(feat_r)
    st = [torch.empty((G, C), device=DEV) for _ in range(4)]
    rc = lib.mft_block_exit_small_forward(ops._p(r1v), ops._p(w2), C * 9 * C, ops._p(sc_r), ops._p(c2), ops._p(out), ops._p(feat), n,
                                          3, 3, C, ipg, ops._p(g2), ops._p(b2), ops._p(gs), ops._p(bs), C, ops._p(st[0]),
                                          ops._p(st[1]), ops._p(st[2]), ops._p(st[3]), 1e-5, ops._stream())
    assert rc == 0
    assert torch.equal(c2, c2_r)
    for a, b in ((st[0], st_r[0]), (st[2], st_r[2])):
        assert float((a - b).abs().max()) < 1e-6
    for a, b in ((st[1], st_r[1]), (st[3], st_r[3])):
        assert float(((a - b) / b).abs().max()) < 1e-5
    assert float((out - out_r).abs().max()) < 2e-5 and float((feat - feat_r).abs().max()) < 1e-5
    # backward half
    dc2 = nhwc(rnd((n, C, 3, 3), 81)).to(DEV)
    dr1 = ops.conv2d_dgrad(dc2, w2, C, 3, 3, 1, imgs_per_group=ipg)
    dx_r, dg_r, db_r = ops.bn_backward(c1_r.view(-1, C), dr1.view(-1, C), C, rows, G, m_r, s_r, g1, relu_out=r1_r,
                                       gb_group_stride=C)
    dx, dg, db = torch.empty_like(dx_r), torch.empty_like(dg_r), torch.empty_like(db_r)
    rc = lib.mft_conv2d_dgrad_bn_backward_small(ops._p(dc2), C, ops._p(w2), ops._p(dx), C, n, 3, 3, C, C, 3, 3, 1, ipg,
                                                C * 9 * C, ops._p(c1_r), ops._p(r1_r), ops._p(m_r), ops._p(s_r), ops._p(g1), C,
                                                ops._p(dg), ops._p(db), ops._stream())
    assert rc == 0
    sc_ = max(float(dx_r.abs().max()), 1e-6)
    assert float((dx - dx_r).abs().max()) <= 2e-5 * sc_
    assert float((dg - dg_r).abs().max()) <= 2e-5 * max(float(dg_r.abs().max()), 1.0)
    assert float((db - db_r).abs().max()) <= 2e-5 * max(float(db_r.abs().max()), 1.0)
    # out-of-domain shapes are refused (callers fall back to the separate launches)
    assert lib.mft_block_entry_small_forward(ops._p(x), Cin, ops._p(w1), C * 9 * Cin, ops._p(wsc), C * Cin, ops._p(c1), ops._p(r1),
                                             ops._p(sc), n, H, H, Cin, C, 2, n, ops._p(g1), ops._p(b1), C, ops._p(m1),
                                             ops._p(s1), 1e-5, ops._stream()) == (-22 if n * 9 > 48 else 0)


@pytest.mark.parametrize("name,Cin,Cout,k,stride,pad,H,ipg", [("trunk.4.C1", 64, 64, 3, 1, 1, 21, 5), ("trunk.5.C1", 64, 128, 3, 2, 1, 21, 5),
                                                             ("trunk.6.C2", 256, 256, 3, 1, 1, 6, 5), ("trunk.5.sc", 64, 128, 1, 2, 0, 21, 3)])
def test_conv_x3_fused_bn_statistics(name, Cin, Cout, k, stride, pad, H, ipg):
    """BatchNorm statistics produced in the bf16x3 convolution epilogue (tiles straddling group boundaries, ragged last tile,
    Chan merge) against float64 statistics of the convolution output, and the same convolution output as the plain launch."""
    from meta_fine_tuning_amd import _lib
    G = 7
    n = G * ipg
    x = nhwc(rnd((n, Cin, H, H), 41) + 0.3).to(DEV)
    w = rnd((Cout, Cin, k, k), 42, scale=(2.0 / (k * k * Cout)) ** 0.5)
    w3 = ops.split_weight_x3(ops.pack_conv_weight(w.to(DEV)))
    OH = (H + 2 * pad - k) // stride + 1
    ref_out = ops.conv2d_x3(x, w3, Cout, k, k, stride, pad)
    out = torch.empty_like(ref_out)
    nws = int(_lib.lib().mft_conv2d_x3_stats_ws_floats(n, H, H, Cout, k, k, stride, pad))
    ws = torch.empty(nws, device=DEV)
    mean, rstd = torch.empty((G, Cout), device=DEV), torch.empty((G, Cout), device=DEV)
    ops.conv2d_x3_bnstats(x, w3, Cout, k, k, stride, pad, ipg, out, ws, mean, rstd)
    assert torch.equal(out, ref_out)
    o = ref_out.double().cpu().view(G, ipg * OH * OH, Cout)
    m_ref = o.mean(1)
    r_ref = 1.0 / torch.sqrt(o.var(1, unbiased=False) + 1e-5)
    assert float((mean.cpu().double() - m_ref).abs().max()) < 2e-6 * max(1.0, float(m_ref.abs().max()))
    assert float(((rstd.cpu().double() - r_ref) / r_ref).abs().max()) < 2e-5


def test_ingest_episode_views_matches_layout_copies():
    """mft_ingest_episode_views (one launch) against the reference's x_a_i assembly (finetune.py:208-233): view 0 twice, then
    views 1.., support images only, NCHW -> NHWC; and view 0 of every image for the final pass."""
    import ctypes
    from meta_fine_tuning_amd import _lib
    n_way, per, ns, H, V = 5, 7, 3, 12, 4
    views = [rnd((n_way, per, 3, H, H), 90 + v).to(DEV) for v in range(V)]
    ptrs = (ctypes.c_void_p * V)(*[v.data_ptr() for v in views])
    npv = n_way * ns
    sup = torch.empty(((V + 1) * npv, H, H, 3), device=DEV)
    allv = torch.empty((n_way * per, H, H, 3), device=DEV)
    assert _lib.lib().mft_ingest_episode_views(ptrs, V, 1, n_way, per, ns, 3, H, H, ops._p(sup), ops._p(allv), ops._stream()) == 0
    ref = torch.cat([views[0][:, :ns].reshape(npv, 3, H, H)] + [v[:, :ns].reshape(npv, 3, H, H) for v in views]).permute(0, 2, 3, 1)
    assert torch.equal(sup, ref.contiguous())
    assert torch.equal(allv, views[0].reshape(n_way * per, 3, H, H).permute(0, 2, 3, 1).contiguous())
    sup1 = torch.empty((npv, H, H, 3), device=DEV)                      # single view, no doubling, no final-pass store
    assert _lib.lib().mft_ingest_episode_views(ptrs, 1, 0, n_way, per, ns, 3, H, H, ops._p(sup1), None, ops._stream()) == 0
    assert torch.equal(sup1, ref[:npv].contiguous())
    assert _lib.lib().mft_ingest_episode_views(ptrs, 33, 0, n_way, per, ns, 3, H, H, ops._p(sup1), None, ops._stream()) == -22


def test_linear_head_sgd_run_matches_torch_sgd():
    """mft_linear_head_sgd_run (all epochs x mini-batches of the set_forward_adaptation head training in one launch, two
    episodes at once, ragged last mini-batch) against torch.optim.SGD(lr .01, momentum .9, dampening .9, weight_decay .001)."""
    from meta_fine_tuning_amd import _lib
    G, S, D, n_way, bs, epochs = 2, 25, 512, 5, 4, 6
    rs = np.random.RandomState(17)
    z = torch.from_numpy(np.abs(rs.standard_normal((G, S, D))).astype(np.float32))
    y = torch.from_numpy(np.stack([np.repeat(np.arange(n_way), S // n_way)[rs.permutation(S)] for _ in range(G)]).astype(np.int64))
    W0 = torch.from_numpy((rs.standard_normal((G, n_way, D)) * 0.04).astype(np.float32))
    b0 = torch.from_numpy((rs.standard_normal((G, n_way)) * 0.01).astype(np.float32))
    steps = [[] for _ in range(G)]
    for g in range(G):
        for _ in range(epochs):
            pm = rs.permutation(S)
            for i in range(0, S, bs):
                ids = pm[i:i + bs]
                steps[g].append(np.concatenate([ids, -np.ones(bs - len(ids), dtype=ids.dtype)]))
    table = torch.from_numpy(np.stack([np.stack(st) for st in steps]).astype(np.int32)).to(DEV)
    W, b = W0.clone().to(DEV), b0.clone().to(DEV)
    z_dev, y_dev = z.to(DEV), y.to(torch.int32).to(DEV)            # named: the launch is asynchronous, temporaries would be freed
    rc = _lib.lib().mft_linear_head_sgd_run(ops._p(z_dev), ops._p(y_dev), ops._p(table), G, S, D, n_way,
                                            table.shape[1], bs, ops._p(W), ops._p(b), 0.01, 0.9, 0.9, 0.001, ops._stream())
    assert rc == 0
    for g in range(G):
        lin = torch.nn.Linear(D, n_way).double()
        lin.weight.data.copy_(W0[g])
        lin.bias.data.copy_(b0[g])
        opt = torch.optim.SGD(lin.parameters(), lr=0.01, momentum=0.9, dampening=0.9, weight_decay=0.001)
        for st in steps[g]:
            ids = torch.from_numpy(st[st >= 0])
            opt.zero_grad()
            F.cross_entropy(lin(z[g][ids].double()), y[g][ids]).backward()
            opt.step()
        dw = float((W[g].cpu().double() - lin.weight.data).abs().max())
        db = float((b[g].cpu().double() - lin.bias.data).abs().max())
        assert dw < 2e-5 and db < 2e-5, (g, dw, db, float((W0[g].double() - lin.weight.data).abs().max()))


@pytest.mark.parametrize("ipg", [5, 4, 1])
def test_wgrad_adam_with_fused_data_gradient(ipg):
    """mft_conv2d_wgrad_adam_dgrad_nhwc + mft_col2im_bn_backward_small (one pass over trunk.7.C2's weights) against
    mft_conv2d_dgrad_nhwc + mft_bn_backward + mft_conv2d_wgrad_adam_nhwc: same weight gradient bit for bit (same reduction
    order), (w, m, v) to the last ulp, the input gradient / dgamma / dbeta to fp32 rounding (different summation order over output channels and taps)."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    G, C, H = 3, 512, 3
    n, rows = G * ipg, ipg * 9
    r1 = torch.relu(nhwc(rnd((n, C, H, H), 101))).to(DEV)
    c1 = nhwc(rnd((n, C, H, H), 102)).to(DEV)
    dc2 = (nhwc(rnd((n, C, H, H), 103)) * 1e-2).to(DEV)
    w0 = torch.stack([ops.pack_conv_weight(rnd((C, C, 3, 3), 104 + g, scale=0.02).to(DEV)) for g in range(G)])
    m0, v0 = (rnd((G, C, 9 * C), 110) * 1e-3).to(DEV), (rnd((G, C, 9 * C), 111).abs() * 1e-6).to(DEV)
    g1 = (rnd((G, C), 112) * 0.2 + 1).to(DEV)
    mean, rstd = ops.bn_stats(c1.view(-1, C), C, rows, G)
    # separate launches
    dr1 = ops.conv2d_dgrad(dc2, w0, C, 3, 3, 1, imgs_per_group=ipg)
    dx_r, dg_r, db_r = ops.bn_backward(c1.view(-1, C), dr1.view(-1, C), C, rows, G, mean, rstd, g1, relu_out=r1.view(-1, C),
                                       gb_group_stride=C)
    w_r, m_r, v_r = w0.clone(), m0.clone(), v0.clone()
    ops.conv2d_wgrad_adam(r1, dc2, w_r, m_r, v_r, C, 3, 3, 1, 1, 7, imgs_per_group=ipg)
    # one pass
    w, m, v = w0.clone(), m0.clone(), v0.clone()
    dxp = torch.empty((G, 9, rows, C), device=DEV)
    assert int(lib.mft_conv2d_wgrad_adam_dgrad_ws_floats(n, H, H, C)) == dxp.numel()
    assert ops.conv2d_wgrad_adam_dgrad(r1, dc2, w, m, v, dxp, 7, ipg)
    dx, dg, db = torch.empty_like(dx_r), torch.empty_like(dg_r), torch.empty_like(db_r)
    assert lib.mft_col2im_bn_backward_small(ops._p(dxp), ops._p(c1), ops._p(r1), ops._p(dx), n, H, H, C, ipg, ops._p(mean),
                                            ops._p(rstd), ops._p(g1), C, ops._p(dg), ops._p(db), ops._stream()) == 0
    # same reduction order -> identical gradient, hence identical first moments; v and w agree to the last ulp or two (the
    # compiler contracts b2*v + (1-b2)*g*g into fused multiply-adds differently in the two kernels)
    # the default weight-gradient launch now uses packed fp32 moment updates and hardware rcp / sqrt: a couple of ulps
    assert float((m - m_r).abs().max()) <= 4e-7 * float(m_r.abs().max())
    assert float(((v - v_r).abs() / v_r.abs().clamp_min(1e-12)).max()) < 1e-6 and float((w - w_r).abs().max()) < 1e-7
    assert float((dx - dx_r).abs().max()) <= 2e-5 * max(float(dx_r.abs().max()), 1e-6)
    assert float((dg - dg_r).abs().max()) <= 2e-5 * max(float(dg_r.abs().max()), 1e-3)
    assert float((db - db_r).abs().max()) <= 2e-5 * max(float(db_r.abs().max()), 1e-3)
    # rows > 64 are outside the kernel's domain
    big = torch.empty((1, 9, 9 * 9, C), device=DEV)
    x9 = torch.zeros((9, H, H, C), device=DEV)
    assert not ops.conv2d_wgrad_adam_dgrad(x9, x9, w[:1], m[:1], v[:1], big, 1, 9)


@pytest.mark.parametrize("name,Cin,Cout,k,stride,pad,H", [("trunk.7.C2", 512, 512, 3, 1, 1, 3), ("trunk.7.C1", 256, 512, 3, 2, 1, 6),
                                                          ("trunk.7.shortcut", 256, 512, 1, 2, 0, 6)])
@pytest.mark.parametrize("ipg", [5, 4, 1])
def test_wgrad_adam_rows_kernel(name, Cin, Cout, k, stride, pad, H, ipg):
    """The stream-shaped 32 x 128 weight-gradient + Adam kernel of the inner loop (<= 64 reduction rows; csrc/conv_igemm.hip
    wgrad_adam_rows_kernel): (i) with exact division / square root it is BIT-IDENTICAL to the 64 x 64 tile kernel (same
    reduction order); (ii) its default epilogue (hardware v_rcp_f32 / v_sqrt_f32, packed moment updates) moves a weight by at
    most a few 1e-9 relative to that; (iii) against torch.optim.Adam on the float64 gradient."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    G = 3
    n = G * ipg
    OH = (H + 2 * pad - k) // stride + 1
    x = rnd((n, Cin, H, H), 201)
    dy = rnd((n, Cout, OH, OH), 202) * 1e-2
    w0 = rnd((G, Cout, Cin, k, k), 203, scale=0.02)
    m0 = rnd((G, Cout, k * k * Cin), 204) * 1e-3
    v0 = rnd((G, Cout, k * k * Cin), 205).abs() * 1e-6
    xg, dyg = nhwc(x).to(DEV), nhwc(dy).to(DEV)
    wpk = torch.stack([ops.pack_conv_weight(w0[g].to(DEV)) for g in range(G)])
    res = {}
    variants = [((9500, 9003), "tile"), ((9501, 9003), "rows_exact"), ((9501, 9007), "rows_fast"), ((9501, 9007, 9600), "rows_fast_padded")]
    if lib.mft_has_experiments():                    # the output-channel-walking form exists in MFT_EXPERIMENTS builds only
        variants += [((9505, 9003), "cowalk_exact"), ((9505, 9007), "cowalk_fast")]
    else:
        assert lib.mft_debug_set_conv_tile(9505) == _lib.MFT_EINVAL and lib.mft_debug_set_conv_tile(9015) == _lib.MFT_EINVAL
    for knobs, tag in variants:
        lib.mft_debug_reset()
        for kn in knobs:
            assert lib.mft_debug_set_conv_tile(kn) == 0
        w, m, v = wpk.clone(), m0.to(DEV).clone(), v0.to(DEV).clone()
        ops.conv2d_wgrad_adam(xg, dyg, w, m, v, Cout, k, k, stride, pad, 7, imgs_per_group=ipg)
        res[tag] = (w, m, v)
    lib.mft_debug_reset()
    for a, b in zip(res["tile"], res["rows_exact"]):
        assert torch.equal(a, b), name
    # the output-channel walk (one workgroup per K tile, im2col rows resident in LDS) and the padded matrix loop change no bit
    for other, base in (("cowalk_exact", "rows_exact"), ("cowalk_fast", "rows_fast"), ("rows_fast_padded", "rows_fast")):
        if other not in res:
            continue
        for a, b in zip(res[other], res[base]):
            assert torch.equal(a, b), (name, other)
    (wf, mf, vf), (we, me, ve) = res["rows_fast"], res["rows_exact"]
    # packed multiply-add contraction: a couple of ulps of the LARGER addend (relative error is unbounded where b1*m and (1-b1)*g cancel)
    assert float((mf - me).abs().max()) <= 4e-7 * float(me.abs().max())
    assert float((vf - ve).abs().max()) <= 4e-7 * float(ve.abs().max())
    assert float((wf - we).abs().max()) < 1e-7, float((wf - we).abs().max())          # ulps of the weight plus ~3e-7 of the update (<= 0.02)
    # torch.optim.Adam on the float64 gradient of the same convolution
    for g in range(G):
        wt = w0[g].double().requires_grad_(True)
        out = F.conv2d(x[g * ipg:(g + 1) * ipg].double(), wt, None, stride, pad)
        out.backward(dy[g * ipg:(g + 1) * ipg].double())
        grad = wt.grad.permute(0, 2, 3, 1).reshape(Cout, -1)                            # packed [Cout][kh][kw][ci]
        b1, b2 = 0.9, 0.999
        mr = b1 * m0[g].double() + (1 - b1) * grad
        vr = b2 * v0[g].double() + (1 - b2) * grad * grad
        wr = wpk[g].cpu().double() - (0.01 / (1 - b1 ** 7)) * mr / (vr.sqrt() / (1 - b2 ** 7) ** 0.5 + 1e-8)
        # (entries with v ~ 1e-9 amplify the fp32-vs-fp64 gradient difference by 1 / sqrt(v))
        assert float((wf[g].cpu().double() - wr).abs().max()) < 2e-5, name
        assert float((mf[g].cpu().double() - mr).abs().max()) < 1e-7


# ---------------------------------------------------------------------- weight gradient + Adam + the next step's convolution

@pytest.mark.parametrize("ipg", [5, 4, 1])
def test_wgrad_adam_next_forward_kernel(ipg):
    """csrc/wgrad_fwd.hip on the three trunk.7 layers (per-episode weights, 84x84 geometry): (i) the gradient and (w, m, v)
    are BIT-IDENTICAL to mft_conv2d_wgrad_adam_nhwc (fast and exact epilogue); (ii) the convolution of the NEXT step's
    activation with the updated weights, and each epilogue -- raw (shortcut), BatchNorm + ReLU (C1), both BatchNorms + add +
    ReLU + average pool (C2) -- against float64 on the updated weights the launch itself produced."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    G = 3
    n = G * ipg
    eps = ops.BN_EPS
    for exact in (0, 1):
        lib.mft_debug_reset()
        lib.mft_wgrad_fwd_set_exact(exact)
        if exact:
            lib.mft_debug_set_conv_tile(9003)
        sc_raw = None
        for name, Cin, Cout, k, stride, pad, H, mode in (("shortcut", 256, 512, 1, 2, 0, 6, ops.WF_RAW), ("C1", 256, 512, 3, 2, 1, 6, ops.WF_ENTRY),
                                                         ("C2", 512, 512, 3, 1, 1, 3, ops.WF_EXIT)):
            OH = (H + 2 * pad - k) // stride + 1
            rows = ipg * OH * OH
            x = rnd((n, Cin, H, H), 301)
            xn = rnd((n, Cin, H, H), 302)
            dy = rnd((n, Cout, OH, OH), 303) * 1e-2
            w0 = rnd((G, Cout, Cin, k, k), 304, scale=0.02)
            m0 = rnd((G, Cout, k * k * Cin), 305) * 1e-3
            v0 = rnd((G, Cout, k * k * Cin), 306).abs() * 1e-6
            gam, bet = 1.0 + 0.1 * rnd((G, Cout), 307), 0.1 * rnd((G, Cout), 308)
            gas, bes = 1.0 + 0.1 * rnd((G, Cout), 309), 0.1 * rnd((G, Cout), 310)
            xg, xng, dyg = nhwc(x).to(DEV), nhwc(xn).to(DEV), nhwc(dy).to(DEV)
            wpk = torch.stack([ops.pack_conv_weight(w0[g].to(DEV)) for g in range(G)])
            # reference launch: gradient + Adam only
            wr, mr, vr = wpk.clone(), m0.to(DEV).clone(), v0.to(DEV).clone()
            dwr = torch.zeros_like(wr)
            ops.conv2d_wgrad_adam(xg, dyg, wr, mr, vr, Cout, k, k, stride, pad, 7, imgs_per_group=ipg, dw=dwr)
            w, m, v = wpk.clone(), m0.to(DEV).clone(), v0.to(DEV).clone()
            dw = torch.zeros_like(w)
            raw = torch.full((n * OH * OH, Cout), float("nan"), device=DEV)
            act = torch.full_like(raw, float("nan"))
            mean, rstd, means, rstds = (torch.full((G, Cout), float("nan"), device=DEV) for _ in range(4))
            pooled = torch.full((n, Cout), float("nan"), device=DEV)
            kw = dict(x_next=xng, mode=mode, raw=raw, dw=dw)
            if mode != ops.WF_RAW:
                kw.update(act=act, gamma=gam.to(DEV), beta=bet.to(DEV), gbs=Cout, mean=mean, rstd=rstd)
            if mode == ops.WF_EXIT:
                kw.update(sc_raw=sc_raw, gamma_s=gas.to(DEV), beta_s=bes.to(DEV), mean_s=means, rstd_s=rstds, pooled=pooled)
            assert ops.wgrad_adam_next_forward(xg, dyg, w, m, v, k, k, stride, pad, 7, ipg, **kw)
            assert torch.equal(dw, dwr) and torch.equal(m, mr) and torch.equal(v, vr) and torch.equal(w, wr), (name, exact)
            # without x_next: the same update, nothing else written
            w2, m2, v2 = wpk.clone(), m0.to(DEV).clone(), v0.to(DEV).clone()
            assert ops.wgrad_adam_next_forward(xg, dyg, w2, m2, v2, k, k, stride, pad, 7, ipg)
            assert torch.equal(w2, wr) and torch.equal(m2, mr) and torch.equal(v2, vr)
            # float64 statement of the next step on the UPDATED weights
            raw_r = torch.empty((G, rows, Cout), dtype=torch.float64)
            for g in range(G):
                wg = ops.unpack_conv_weight(w[g].contiguous(), (Cout, Cin, k, k)).cpu().double()
                o = F.conv2d(xn[g * ipg:(g + 1) * ipg].double(), wg, None, stride, pad)
                raw_r[g] = o.permute(0, 2, 3, 1).reshape(rows, Cout)
            got = raw.view(G, rows, Cout).cpu().double()
            scale = float(raw_r.abs().max())
            assert float((got - raw_r).abs().max()) <= 3e-6 * scale, (name, float((got - raw_r).abs().max()), scale)
            if mode == ops.WF_RAW:
                sc_raw = raw.clone()
                sc_raw_r = raw_r
                continue
            mu = raw_r.mean(1, keepdim=True)
            rs = 1.0 / (raw_r.var(1, unbiased=False, keepdim=True) + eps).sqrt()
            assert float((mean.cpu().double() - mu[:, 0]).abs().max()) <= 3e-6 * scale
            assert float((rstd.cpu().double() / rs[:, 0] - 1).abs().max()) <= 2e-5
            y = (raw_r - mu) * rs * gam.double()[:, None] + bet.double()[:, None]
            if mode == ops.WF_EXIT:
                mus = sc_raw_r.mean(1, keepdim=True)
                rss = 1.0 / (sc_raw_r.var(1, unbiased=False, keepdim=True) + eps).sqrt()
                assert float((means.cpu().double() - mus[:, 0]).abs().max()) <= 3e-6 * float(sc_raw_r.abs().max())
                assert float((rstds.cpu().double() / rss[:, 0] - 1).abs().max()) <= 2e-5
                y = y + (sc_raw_r - mus) * rss * gas.double()[:, None] + bes.double()[:, None]
            y = y.clamp_min(0)
            assert float((act.view(G, rows, Cout).cpu().double() - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1.0), name
            if mode == ops.WF_EXIT:
                pr = y.view(G, ipg, OH * OH, Cout).mean(2).reshape(n, Cout)
                assert float((pooled.cpu().double() - pr).abs().max()) <= 2e-5 * max(float(pr.abs().max()), 1.0)
    lib.mft_wgrad_fwd_set_exact(0)
    lib.mft_debug_reset()
    # outside the domain: more than 48 output pixels per episode
    xb = torch.zeros((6, 6, 6, 256), device=DEV)
    dyb = torch.zeros((6, 3, 3, 512), device=DEV)
    wb = torch.zeros((1, 512, 2304), device=DEV)
    assert not ops.wgrad_adam_next_forward(xb, dyb, wb, wb.clone(), wb.clone(), 3, 3, 2, 1, 1, 6)


@pytest.mark.parametrize("G", [16, 24, 12])
def test_wgrad_adam_next_forward_xcd_order_is_bit_identical(G):
    """The fused launch places all workgroups of an episode on one XCD (mft_wgrad_fwd_set_xcd, default on when the episode count is
    a multiple of 8): placement only -- w, m, v, the gradient and every output of the next step's forward equal the natural
    workgroup order bit for bit (16 / 24 episodes: remapped; 12: not a multiple of 8, natural order either way)."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    ipg, Cin, Cout, k, stride, pad, H = 5, 256, 512, 3, 2, 1, 6
    n = G * ipg
    OH = (H + 2 * pad - k) // stride + 1
    xg, xng = nhwc(rnd((n, Cin, H, H), 401)).to(DEV), nhwc(rnd((n, Cin, H, H), 402)).to(DEV)
    dyg = nhwc(rnd((n, Cout, OH, OH), 403) * 1e-2).to(DEV)
    w0 = rnd((G, Cout, k * k * Cin), 404, scale=0.02).to(DEV)
    m0, v0 = (rnd((G, Cout, k * k * Cin), 405) * 1e-3).to(DEV), (rnd((G, Cout, k * k * Cin), 406).abs() * 1e-6).to(DEV)
    gam, bet = (1.0 + 0.1 * rnd((G, Cout), 407)).to(DEV), (0.1 * rnd((G, Cout), 408)).to(DEV)
    outs = []
    try:
        for on in (1, 0):
            lib.mft_wgrad_fwd_set_xcd(on)
            w, m, v = w0.clone(), m0.clone(), v0.clone()
            dw = torch.zeros_like(w)
            raw = torch.full((n * OH * OH, Cout), float("nan"), device=DEV)
            act = torch.full_like(raw, float("nan"))
            mean, rstd = torch.full((G, Cout), float("nan"), device=DEV), torch.full((G, Cout), float("nan"), device=DEV)
            assert ops.wgrad_adam_next_forward(xg, dyg, w, m, v, k, k, stride, pad, 3, ipg, x_next=xng, mode=ops.WF_ENTRY, raw=raw, act=act,
                                               gamma=gam, beta=bet, gbs=Cout, mean=mean, rstd=rstd, dw=dw)
            outs.append((w, m, v, dw, raw, act, mean, rstd))
    finally:
        lib.mft_wgrad_fwd_set_xcd(1)
    for a, b in zip(*outs):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    assert not torch.equal(outs[0][0], w0)


@pytest.mark.parametrize("Cin,Cout,H,ipg,G", [(64, 64, 21, 5, 6), (128, 128, 11, 5, 7), (64, 128, 11, 3, 5), (256, 256, 6, 5, 4)])
def test_conv_x3_loader_side_batchnorm_is_bit_identical(Cin, Cout, H, ipg, G):
    """SimpleBlock's C1 -> BN1 -> ReLU -> C2 (backbone.py:251-256) with BN1 folded into C2's loader
    (mft_conv2d_nhwc_x3_bnin_bnstats: statistics merged from C1's partials in the workgroup prologue) against the separate
    launches finalize -> mft_bn_apply -> mft_conv2d_nhwc_x3_bnstats: same arithmetic, so the convolution output and the
    statistics behind it must match bit for bit (tiles straddling groups, ragged last tile, image-row boundaries, negative
    gammas).  A 512-channel table does not fit beside the tile at three workgroups per CU: refused (MFT_EINVAL -> None)."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    n = G * ipg
    x = nhwc(rnd((n, 64, H, H), 71) + 0.2).to(DEV)
    w1 = ops.split_weight_x3(ops.pack_conv_weight(rnd((Cin, 64, 3, 3), 72, scale=(2.0 / (9 * Cin)) ** 0.5).to(DEV)))
    w2 = ops.split_weight_x3(ops.pack_conv_weight(rnd((Cout, Cin, 3, 3), 73, scale=(2.0 / (9 * Cout)) ** 0.5).to(DEV)))
    g1 = (rnd((Cin,), 74) * 0.5 + 0.3).to(DEV)                        # some negative gammas
    b1 = (rnd((Cin,), 75) * 0.2).to(DEV)
    c1 = torch.empty((n, H, H, Cin), device=DEV)
    ws1 = torch.empty(int(lib.mft_conv2d_x3_stats_ws_floats(n, H, H, Cin, 3, 3, 1, 1)), device=DEV)
    m1, s1 = torch.empty((G, Cin), device=DEV), torch.empty((G, Cin), device=DEV)
    ops.conv2d_x3_bnstats(x, w1, Cin, 3, 3, 1, 1, ipg, c1, ws1, m1, s1)
    rows = ipg * H * H
    r1 = ops.bn_apply(c1.view(-1, Cin), Cin, rows, G, m1, s1, g1, b1, act=ops.ACT_RELU, fma_affine=True).view(n, H, H, Cin)
    ws2 = torch.empty(int(lib.mft_conv2d_x3_stats_ws_floats(n, H, H, Cout, 3, 3, 1, 1)), device=DEV)
    ref = torch.empty((n, H, H, Cout), device=DEV)
    m2, s2 = torch.empty((G, Cout), device=DEV), torch.empty((G, Cout), device=DEV)
    ops.conv2d_x3_bnstats(r1, w2, Cout, 3, 3, 1, 1, ipg, ref, ws2, m2, s2)
    # the partials-only form of the first convolution leaves the same partials and launches no finalize
    c1b, ws1b = torch.empty_like(c1), torch.empty_like(ws1)
    assert ops.conv2d_x3_bnstats(x, w1, Cin, 3, 3, 1, 1, ipg, c1b, ws1b, None, None) is not None
    n_part = ((n * H * H + 127) // 128) * 2 * Cin * 2
    assert torch.equal(c1b, c1)
    out = torch.zeros_like(ref)
    ws2b = torch.empty_like(ws2)
    m2b, s2b = torch.empty_like(m2), torch.empty_like(s2)
    r = ops.conv2d_x3_bnin_bnstats(c1b, ws1b, g1, b1, w2, Cout, ipg, out, ws2b, m2b, s2b)
    # (256 channels on 6x6 maps were refused while the staged image carried a zero row per image-row boundary: 152 rows; with one
    #  zero row -- 131 rows -- the (scale, shift) table fits beside the tile at three workgroups per CU.)  The stand-alone apply
    #  from partials that a refusing shape falls back to is checked against the plain apply either way:
    r1b = ops.bn_apply_x3ws(c1b.view(-1, Cin), Cin, rows, G, ws1b, g1, b1, torch.empty((n * H * H, Cin), device=DEV), act=ops.ACT_RELU)
    assert torch.equal(r1b.view_as(r1), r1)
    assert r is not None
    assert torch.equal(out, ref)
    assert torch.equal(m2b, m2) and torch.equal(s2b, s2)
    # against float64: relu(BN(c1)) convolved in double
    c1d = c1.double().cpu().view(G, rows, Cin)
    mu, var = c1d.mean(1, keepdim=True), c1d.var(1, unbiased=False, keepdim=True)
    r1d = torch.relu((c1d - mu) / torch.sqrt(var + 1e-5) * g1.double().cpu() + b1.double().cpu()).view(n, H, H, Cin)
    wd = rnd((Cout, Cin, 3, 3), 73, scale=(2.0 / (9 * Cout)) ** 0.5).double()
    od = torch.nn.functional.conv2d(r1d.permute(0, 3, 1, 2), wd, padding=1).permute(0, 2, 3, 1)
    assert float((out.double().cpu() - od).abs().max()) < 2e-5 * max(1.0, float(od.abs().max()))
    del n_part
    if Cin == 256:                                                       # outside the domain: the caller runs apply + plain convolution
        xb = torch.zeros((8, 6, 6, 512), device=DEV)                      # groups of 4 images = 144 rows: inside the row domain
        wsb = torch.zeros(int(lib.mft_conv2d_x3_stats_ws_floats(8, 6, 6, 512, 3, 3, 1, 1)), device=DEV)
        wb = ops.split_weight_x3(ops.pack_conv_weight(torch.zeros((64, 512, 3, 3), device=DEV)))
        gb = torch.ones(512, device=DEV)
        assert ops.conv2d_x3_bnin_bnstats(xb, wsb, gb, gb, wb, 64, 4, torch.empty((8, 6, 6, 64), device=DEV),
                                          torch.empty(int(lib.mft_conv2d_x3_stats_ws_floats(8, 6, 6, 64, 3, 3, 1, 1)), device=DEV)) is None


@pytest.mark.parametrize("C,H,ipg,G,res", [(64, 21, 5, 6, "identity"), (128, 11, 5, 7, "bn"), (256, 6, 5, 5, "bn"), (64, 21, 3, 4, "none")])
def test_bn_apply_from_partials_is_bit_identical(C, H, ipg, G, res):
    """mft_bn_apply_x3ws (statistics of the main and the residual BatchNorm merged from convolution partials inside the apply
    launch) against finalize + finalize + mft_bn_apply: output and merged statistics bit for bit."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    n = G * ipg
    x = nhwc(rnd((n, 64, H, H), 81) + 0.1).to(DEV)
    rows = ipg * H * H

    def conv(seed, k, pad, full):
        w3 = ops.split_weight_x3(ops.pack_conv_weight(rnd((C, 64, k, k), seed, scale=(2.0 / (k * k * C)) ** 0.5).to(DEV)))
        o = torch.empty((n, H, H, C), device=DEV)
        ws = torch.empty(int(lib.mft_conv2d_x3_stats_ws_floats(n, H, H, C, k, k, 1, pad)), device=DEV)
        m, s = (torch.empty((G, C), device=DEV), torch.empty((G, C), device=DEV)) if full else (None, None)
        assert ops.conv2d_x3_bnstats(x, w3, C, k, k, 1, pad, ipg, o, ws, m, s) is not None
        return o, ws, m, s

    c2, ws2, m2, s2 = conv(82, 3, 1, True)
    g2, b2 = (rnd((C,), 83) * 0.5 + 0.4).to(DEV), (rnd((C,), 84) * 0.1).to(DEV)
    kw_ref, kw_new = {}, {}
    if res == "bn":
        sc, wss, ms, ss = conv(85, 1, 0, True)
        gs, bs = (rnd((C,), 86) * 0.5 + 0.4).to(DEV), (rnd((C,), 87) * 0.1).to(DEV)
        kw_ref = dict(res=sc.view(-1, C), res_bn=(ms, ss, gs, bs))
        rm, rr = torch.empty((G, C), device=DEV), torch.empty((G, C), device=DEV)
        kw_new = dict(res=sc.view(-1, C), res_ws=wss, res_gamma=gs, res_beta=bs, res_stats=(rm, rr))
    elif res == "identity":
        assert C == 64
        kw_ref = kw_new = dict(res=x.view(-1, 64))
    ref = ops.bn_apply(c2.view(-1, C), C, rows, G, m2, s2, g2, b2, act=ops.ACT_RELU, fma_affine=True, **kw_ref)
    mo, so = torch.empty((G, C), device=DEV), torch.empty((G, C), device=DEV)
    out = ops.bn_apply_x3ws(c2.view(-1, C), C, rows, G, ws2, g2, b2, torch.empty_like(ref), act=ops.ACT_RELU, stats=(mo, so), **kw_new)
    assert torch.equal(out, ref)
    assert torch.equal(mo, m2) and torch.equal(so, s2)
    if res == "bn":
        assert torch.equal(rm, ms) and torch.equal(rr, ss)
    # the apply itself against float64
    c2d = c2.double().cpu().view(G, rows, C)
    yd = (c2d - c2d.mean(1, keepdim=True)) / torch.sqrt(c2d.var(1, unbiased=False, keepdim=True) + 1e-5) * g2.double().cpu() + b2.double().cpu()
    if res == "identity":
        yd = yd + x.double().cpu().view(G, rows, C)
    if res != "bn":
        assert float((out.double().cpu().view(G, rows, C) - torch.relu(yd)).abs().max()) < 1e-5 * max(1.0, float(yd.abs().max()))
    assert lib.mft_bn_apply_x3ws(ops._p(c2), C, ops._p(out), C, C, 64, G, ops._p(ws2), ops._p(g2), ops._p(b2), None, 0, None, None, None,
                                 1, 0.0, 1e-5, None, None, None, None, ops._stream()) == -22        # groups below one 128-row tile


@pytest.mark.parametrize("n,H,Cin,Cout,k,stride,pad", [(105, 3, 512, 512, 3, 1, 1), (105, 6, 256, 512, 3, 2, 1), (105, 11, 128, 256, 3, 2, 1),
                                                       (105, 6, 256, 256, 3, 1, 1), (21, 6, 256, 512, 1, 2, 0), (9, 21, 64, 128, 3, 2, 1),
                                                       (7, 11, 128, 256, 1, 2, 0)])
def test_conv_k_sliced_forms_match_plain_and_float64(n, H, Cin, Cout, k, stride, pad):
    """One 105-image meta-training episode leaves the deep layers with 60-240 output tiles and 72-144 K-steps each: the K-sliced
    forward / data-gradient launches (partials summed in slice order) against the plain launches and against float64, and the
    adaptive split-M weight gradient against float64."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    x = nhwc(rnd((n, Cin, H, H), 91)).to(DEV)
    w = rnd((Cout, Cin, k, k), 92, scale=(2.0 / (k * k * Cin)) ** 0.5)
    b = rnd((Cout,), 93).to(DEV)
    wp = ops.pack_conv_weight(w.to(DEV))
    OH = (H + 2 * pad - k) // stride + 1
    sliced = int(lib.mft_conv_ksplit_ws_floats(n * OH * OH, Cout, wp.shape[-1])) > 0
    assert sliced == (k == 3)                                     # the 1x1 shortcuts (K <= 256: 8 K-steps) are not sliced
    out = ops.conv2d(x, wp, Cout, k, k, stride, pad, bias=b)                 # takes the sliced form where it applies
    plain = torch.empty_like(out)
    assert lib.mft_conv2d_nhwc(ops._p(x), Cin, ops._p(wp), ops._p(b), ops._p(plain), Cout, n, H, H, Cin, Cout, k, k, stride, pad, 0, 0,
                               ops._stream()) == 0
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double(), b.double().cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    assert float((out.double().cpu() - ref).abs().max()) < 1e-5 * sc
    assert float((out - plain).abs().max()) < 1e-5 * sc
    dy = nhwc(rnd((n, Cout, OH, OH), 94)).to(DEV)
    dx = ops.conv2d_dgrad(dy, wp, Cin, k, k, pad, stride=stride, in_hw=(H, H))
    dxp = torch.empty_like(dx)
    lib.mft_debug_set_conv_tile(9800)                             # all nine taps with zero rows instead of the parity-class walk (stride 2)
    try:
        assert lib.mft_conv2d_dgrad_nhwc(ops._p(dy), Cout, ops._p(wp), ops._p(dxp), Cin, n, H, H, Cin, Cout, k, k, stride, pad, 0, 0,
                                         ops._stream()) == 0
    finally:
        lib.mft_debug_reset()
    xd = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    o = F.conv2d(xd, wd, stride=stride, padding=pad)
    gx, gw = torch.autograd.grad(o, [xd, wd], dy.double().cpu().permute(0, 3, 1, 2))
    sx = float(gx.abs().max())
    assert float((dx.double().cpu() - gx.permute(0, 2, 3, 1)).abs().max()) < 1e-5 * sx
    assert float((dx - dxp).abs().max()) < 1e-5 * sx
    dw = ops.unpack_conv_weight(ops.conv2d_wgrad(x, dy, Cout, k, k, stride, pad)[0], (Cout, Cin, k, k))
    assert float((dw.double().cpu() - gw).abs().max()) < 2e-5 * float(gw.abs().max())


@pytest.mark.parametrize("rows,C,ld", [(7440, 192, 192), (7440, 96, 96), (7440, 32, 32), (2048, 64, 96), (480, 64, 64), (105, 128, 128), (33, 5, 8)])
def test_colsum_matches_float64_and_is_deterministic(rows, C, ld):
    """mft_colsum (bias gradients of nn.Linear / 1x1 nn.Conv2d, out[c] = sum_r x[r][c]): partial + final launches, float4 and
    scalar forms, against float64; reruns are bit-identical (fixed summation order).  (A one-launch form -- 1024 row lanes + an
    LDS tree -- was measured in round 5: 3.69-3.70 ms per meta-training step against 3.65-3.67 with the pair; not kept.)"""
    from meta_fine_tuning_amd import functional_bwd as FB
    x = rnd((rows, ld), 61).to(DEV)
    got = FB.colsum(x, C)
    ref = x.double().cpu()[:, :C].sum(0)
    assert got.shape == (C,)
    assert float((got.double().cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(got, FB.colsum(x, C))


@pytest.mark.parametrize("n,H,cin,cout", [(105, 21, 64, 64), (105, 11, 128, 128), (7, 11, 64, 128)])
def test_split_plan_planes_and_data_gradient_on_the_split_kernels(n, H, cin, cout):
    """ops.SplitPlan (mft_split_bf16x3_multi, the per-step plane refresh of the meta-training layers that run on the bf16x3 kernels):
    its planes equal mft_split_bf16x3's bit for bit -- of the packed weights, and with ``transposed`` of the data-gradient operand
    (mft_pack_dgrad's tap-flipped channel-swapped matrix) --, a refresh after an in-place weight update tracks the new values, and
    the stride-1 data gradient computed as a convolution over the transposed planes agrees with the fp32 data-gradient launch and
    with float64."""
    w = rnd((cout, cin, 3, 3), 71, scale=(2.0 / (9 * cin)) ** 0.5).to(DEV)
    wp = ops.pack_conv_weight(w)
    plan = ops.SplitPlan()
    pf = plan.add(wp, cout, cin, 9, False)
    pt = plan.add(wp, cout, cin, 9, True) if cin % 64 == 0 else None
    assert bool((pf == ops.split_weight_x3(wp)).all())
    if pt is not None:
        wt = ops.pack_dgrad_weight(wp, cout, cin, 3, 3)[0].contiguous()
        assert pt.shape == (3, cin, 9 * cout) and bool((pt == ops.split_weight_x3(wt)).all())
    wp.mul_(1.5).add_(0.01)                                  # optimizer.step() + repack: same buffer, new values
    plan.run()
    assert bool((pf == ops.split_weight_x3(wp)).all())
    if pt is None:
        return
    assert bool((pt == ops.split_weight_x3(ops.pack_dgrad_weight(wp, cout, cin, 3, 3)[0].contiguous())).all())
    dy = nhwc(rnd((n, cout, H, H), 72)).to(DEV)
    dx = ops.conv2d_x3(dy, pt, cin, 3, 3, 1, 1)
    dx32 = ops.conv2d_dgrad(dy, wp, cin, 3, 3, 1)
    w_now = ops.unpack_conv_weight(wp, (cout, cin, 3, 3))
    ref = torch.nn.grad.conv2d_input((n, cin, H, H), w_now.double().cpu(), nchw(dy.cpu()).double(), stride=1, padding=1)
    sc = float(ref.abs().max())
    assert float((nchw(dx.cpu()).double() - ref).abs().max()) < 1e-5 * sc
    assert float((dx - dx32).abs().max()) < 1e-5 * sc


@pytest.mark.parametrize("n,H,Cin,Cout,k,stride,pad", [(105, 21, 64, 64, 3, 1, 1), (105, 6, 256, 512, 3, 2, 1), (21, 11, 128, 256, 1, 2, 0),
                                                       (4, 84, 3, 64, 7, 2, 3), (3, 6, 256, 256, 3, 1, 1)])
def test_conv_wgrad_written_as_oihw(n, H, Cin, Cout, k, stride, pad):
    """mft_conv2d_wgrad_oihw (the split-M partial sum writes Conv2d.weight.grad's layout) against wgrad + mft_unpack_oihw (bit for
    bit: same partials, same summation order) and against float64."""
    x = nhwc(rnd((n, Cin, H, H), 95)).to(DEV)
    OH = (H + 2 * pad - k) // stride + 1
    dy = nhwc(rnd((n, Cout, OH, OH), 96)).to(DEV)
    from meta_fine_tuning_amd import _lib
    if Cin == 3:                    # (the stem's oihw gradient has its own strip kernel by default: the generic form for the bit-for-bit part)
        assert _lib.lib().mft_debug_set_conv_tile(2100) == 0
    got = ops.conv2d_wgrad_oihw(x, dy, Cout, k, k, stride, pad)
    ref = ops.unpack_conv_weight(ops.conv2d_wgrad(x, dy, Cout, k, k, stride, pad)[0], (Cout, Cin, k, k))
    assert got.shape == (Cout, Cin, k, k) and torch.equal(got, ref)
    _lib.lib().mft_debug_reset()
    xd = x.double().cpu().permute(0, 3, 1, 2)
    wd = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    gw, = torch.autograd.grad(F.conv2d(xd, wd, stride=stride, padding=pad), wd, dy.double().cpu().permute(0, 3, 1, 2))
    assert float((got.double().cpu() - gw).abs().max()) < 2e-5 * float(gw.abs().max())


@pytest.mark.parametrize("n,H", [(105, 84), (21, 84), (4, 224), (3, 50), (420, 84)])
def test_stem_wgrad_strip_kernel(n, H):
    """trunk.0's weight gradient in torch's layout by the strip kernel (csrc/conv_igemm.hip: dy and seven image rows per strip of one
    output row in LDS, the (kw, ci) run of an im2col row read at offset 6 m + j) against float64 and against the generic gather
    kernel (test hook 2100) on the same inputs; run twice -> bit-identical.  224 x 224: two strips per output row; 50 x 50: odd strip."""
    from meta_fine_tuning_amd import _lib
    x = ops.nchw_to_nhwc(rnd((n, 3, H, H), 131).to(DEV))
    OH = (H + 6 - 7) // 2 + 1
    dy = nhwc(rnd((n, 64, OH, OH), 132)).to(DEV)
    got = ops.conv2d_wgrad_oihw(x, dy, 64, 7, 7, 2, 3)
    got2 = ops.conv2d_wgrad_oihw(x, dy, 64, 7, 7, 2, 3)
    assert _lib.lib().mft_debug_set_conv_tile(2100) == 0
    gen = ops.conv2d_wgrad_oihw(x, dy, 64, 7, 7, 2, 3)
    _lib.lib().mft_debug_reset()
    assert got.shape == (64, 3, 7, 7) and torch.equal(got, got2)
    gw = torch.nn.grad.conv2d_weight(x.double().cpu().permute(0, 3, 1, 2), (64, 3, 7, 7), dy.double().cpu().permute(0, 3, 1, 2), stride=2, padding=3)
    sc = float(gw.abs().max())
    assert float((got.double().cpu() - gw).abs().max()) < 3e-5 * sc, float((got.double().cpu() - gw).abs().max()) / sc
    assert float((got - gen).abs().max()) < 3e-5 * sc


@pytest.mark.parametrize("n,H,C", [(5, 42, 64), (3, 13, 32), (2, 8, 6)])
def test_maxpool_relu_backward_matches_autograd(n, H, C):
    """BatchNorm -> ReLU -> MaxPool2d(3, 2, 1) of the stem (backbone.py:295-297) with the argmax recorded, and its gradient w.r.t.
    the BatchNorm output (four channels per thread where C % 4 == 0, scalar otherwise) against torch autograd."""
    from meta_fine_tuning_amd import _lib
    lib = _lib.lib()
    x = nhwc(rnd((n, C, H, H), 97)).to(DEV)
    PH = (H + 2 - 3) // 2 + 1
    y = torch.empty((n, PH, PH, C), device=DEV)
    arg = torch.empty((n, PH, PH, C), device=DEV, dtype=torch.uint8)
    zeros, ones = torch.zeros((1, C), device=DEV), torch.ones((1, C), device=DEV)
    rstd = torch.full((1, C), float(1.0 / np.sqrt(1.0 + 1e-5)), device=DEV)
    assert lib.mft_bn_relu_maxpool_arg(ops._p(x), ops._p(y), ops._p(arg), n, H, H, C, n, ops._p(zeros), ops._p(ones), ops._p(ones),
                                       ops._p(zeros), ops._stream()) == 0
    del rstd
    xa = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    ya = F.max_pool2d(F.relu(xa), 3, 2, 1)
    assert float((y.double().cpu().permute(0, 3, 1, 2) - ya).abs().max()) < 1e-6
    dy = nhwc(rnd((n, C, PH, PH), 98)).to(DEV)
    gx, = torch.autograd.grad(ya, xa, dy.double().cpu().permute(0, 3, 1, 2))
    dx = torch.empty_like(x)
    assert lib.mft_maxpool_relu_backward(ops._p(dy), ops._p(arg), ops._p(y), ops._p(dx), n, H, H, C, ops._stream()) == 0
    assert float((dx.double().cpu().permute(0, 3, 1, 2) - gx).abs().max()) < 1e-6


def test_bn_running_ema_matches_sequential_torch_updates():
    """mft_bn_running_ema: running_mean / running_var after a sequence of train-mode BatchNorm forwards whose batch statistics came
    out of two grouped launches (full and ragged mini-batches), against F.batch_norm(training=True) applied step by step."""
    from meta_fine_tuning_amd import _lib
    C, rows_a, rows_b = 96, 36, 9
    rs = np.random.RandomState(5)
    xa = torch.from_numpy(rs.standard_normal((6, rows_a, C)).astype(np.float32) * 1.5 + 0.3)
    xb = torch.from_numpy(rs.standard_normal((3, rows_b, C)).astype(np.float32) * 0.7 - 0.2)
    order = [(0, 0), (0, 1), (1, 0), (0, 2), (0, 3), (1, 1), (0, 4), (1, 2), (0, 5)]
    rm, rv = torch.zeros(C), torch.ones(C)
    for kind, g in order:                                   # torch's own running-statistics update, one forward at a time
        x = (xa, xb)[kind][g]
        F.batch_norm(x, rm, rv, None, None, True, 0.1, 1e-5)
    ma, ra = ops.bn_stats(xa.view(-1, C).to(DEV), C, rows_a, 6)
    mb, rb = ops.bn_stats(xb.view(-1, C).to(DEV), C, rows_b, 3)
    od = torch.tensor([(k << 24) | g for k, g in order], dtype=torch.int32, device=DEV)
    grm, grv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    assert _lib.lib().mft_bn_running_ema(ops._p(ma), ops._p(ra), rows_a, ops._p(mb), ops._p(rb), rows_b, ops._p(od), len(order), C, 1e-5, 0.1,
                                         ops._p(grm), ops._p(grv), ops._stream()) == 0
    np.testing.assert_allclose(grm.cpu().numpy(), rm.numpy(), atol=2e-6)
    np.testing.assert_allclose(grv.cpu().numpy(), rv.numpy(), rtol=2e-5)
    assert _lib.lib().mft_bn_running_ema(ops._p(ma), ops._p(ra), 0, None, None, 1, ops._p(od), len(order), C, 1e-5, 0.1, ops._p(grm),
                                         ops._p(grv), ops._stream()) == -22


@pytest.mark.parametrize("G,ipg,H,Cin,Cout,k,stride,pad", [(4, 5, 6, 256, 512, 3, 2, 1), (3, 4, 3, 512, 512, 3, 1, 1), (8, 5, 6, 256, 512, 1, 2, 0),
                                                           (2, 1, 3, 512, 512, 3, 1, 1)])
def test_grouped_k_sliced_convolutions_match_float64(G, ipg, H, Cin, Cout, k, stride, pad, monkeypatch):
    """A few episodes in lockstep (2-8 per-episode weight sets): forward and data gradient on the K-sliced implicit GEMM with
    grid.y = episode, against float64 per episode and against the weight-streaming kernels large batches use."""
    from meta_fine_tuning_amd import _lib
    n = G * ipg
    x = nhwc(rnd((n, Cin, H, H), 111)).to(DEV)
    ws_ = [rnd((Cout, Cin, k, k), 120 + g, scale=(2.0 / (k * k * Cin)) ** 0.5) for g in range(G)]
    wp = torch.stack([ops.pack_conv_weight(w.to(DEV)) for w in ws_])
    OH = (H + 2 * pad - k) // stride + 1
    assert (int(_lib.lib().mft_conv_ksplit_grouped_ws_floats(ipg * OH * OH, Cout, wp.shape[-1], G)) > 0) == (k == 3)
    out = ops.conv2d(x, wp, Cout, k, k, stride, pad, imgs_per_group=ipg)
    monkeypatch.setattr(ops, "SMALL_GROUPS", 0)
    big = ops.conv2d(x, wp, Cout, k, k, stride, pad, imgs_per_group=ipg)            # weight-streaming bf16x3 kernels
    monkeypatch.undo()
    xd = x.double().cpu().permute(0, 3, 1, 2)
    ref = torch.cat([F.conv2d(xd[g * ipg:(g + 1) * ipg], ws_[g].double(), stride=stride, padding=pad) for g in range(G)]).permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    assert float((out.double().cpu() - ref).abs().max()) < 1e-5 * sc
    assert float((out - big).abs().max()) < 2e-5 * sc
    if stride == 1:
        dy = nhwc(rnd((n, Cout, OH, OH), 112)).to(DEV)
        dx = ops.conv2d_dgrad(dy, wp, Cin, k, k, pad, imgs_per_group=ipg)
        dyd = dy.double().cpu().permute(0, 3, 1, 2)
        gref = torch.cat([torch.nn.grad.conv2d_input((ipg, Cin, H, H), ws_[g].double(), dyd[g * ipg:(g + 1) * ipg], padding=pad)
                          for g in range(G)]).permute(0, 2, 3, 1)
        assert float((dx.double().cpu() - gref).abs().max()) < 1e-5 * float(gref.abs().max())


def test_launch_timer_brackets_launchers_on_their_own_streams():
    """_lib.LaunchTimer (bench.py's roofline source): every launcher call inside the block is timed by a pair of HIP events on the
    stream it enqueues on -- a side stream included, which torch.cuda.Event on the current stream would not see --, refused calls
    (MFT_EINVAL) are not counted, the C-ABI arguments are kept on request, and results are unchanged."""
    from meta_fine_tuning_amd import _lib
    x = torch.randn(64, 21, 21, 64, device=DEV)
    w = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=DEV) * 0.05)
    ref = ops.conv2d(x, w, 64, 3, 3, 1, 1).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with _lib.LaunchTimer(keep_args=True) as lt:
        assert _lib.lib() is lt
        a = ops.conv2d(x, w, 64, 3, 3, 1, 1)
        with torch.cuda.stream(side):
            b = ops.conv2d(x, w, 64, 3, 3, 1, 1)
            c = ops.softmax_rows(torch.randn(128, 5, device=DEV))
        rc = _lib.lib().mft_stream_probe(ops._p(x), ops._p(x), ops._p(x), 100, ops._stream())      # n not a multiple of 1024: refused
        assert rc == _lib.MFT_EINVAL
        torch.cuda.synchronize()
        calls = lt.collect(calls=True)
        lt.close()
    assert not isinstance(_lib.lib(), _lib.LaunchTimer)
    names = [n for n, _, _ in calls]
    assert len(names) == 3 and names[0] == names[1] and names[0] in ("mft_conv2d_nhwc", "mft_conv2d_nhwc_ksplit") and names[2] == "mft_softmax_rows"
    assert all(0.0 < ms < 50.0 for _, ms, _ in calls)
    args = calls[0][2]
    assert (args[6], args[7], args[8], args[9], args[10], args[11], args[12]) == (64, 21, 21, 64, 64, 3, 3)      # n, H, W, Cin, Cout, KH, KW
    assert torch.equal(a, ref) and torch.equal(b, ref) and c.shape == (128, 5)
    only = _lib.LaunchTimer(only=lambda n: n == "mft_softmax_rows")
    with only:
        ops.conv2d(x, w, 64, 3, 3, 1, 1)
        ops.softmax_rows(torch.randn(8, 5, device=DEV))
        torch.cuda.synchronize()
        got = only.collect()
        only.close()
    assert list(got) == ["mft_softmax_rows"] and len(got["mft_softmax_rows"]) == 1
    # a block that raises: the library handle is restored and the pending event pairs are released, not leaked
    boom = _lib.LaunchTimer()
    with pytest.raises(ZeroDivisionError):
        with boom:
            ops.conv2d(x, w, 64, 3, 3, 1, 1)
            assert len(boom._pairs) == 1
            1 / 0
    assert not isinstance(_lib.lib(), _lib.LaunchTimer) and boom._pairs == [] and boom._free == []
    torch.cuda.synchronize()


@pytest.mark.parametrize("n", [1, 7, 600])
def test_stem_cache_fused_fill_matches_the_three_launches(n, monkeypatch):
    """mft_stem_cache_fill (csrc/stem.hip: stem_cache_kernel -- trunk.0 + per-image moments + per-window (max, min) in ONE launch, the
    full-resolution output never written) against the three launches it replaces (mft_conv2d_nhwc + mft_bn_image_moments +
    mft_pool_window_minmax): window extrema EQUAL (max / min of the same convolution results), moments to fp32 rounding, and both
    against a float64 statement of the op."""
    from meta_fine_tuning_amd import functional as Fn
    sd = synthetic.resnet10_state_dict(seed=5)
    W = Fn.ResNet10Weights(sd, DEV)
    x = torch.randn(n, 84, 84, 3, device=DEV) * 1.7 + 0.3
    monkeypatch.setenv("MFT_STEM_FUSED_FILL", "1")
    a = Fn.StemCache(W, n, 84, DEV, pooled=True)
    assert a.fused_fill and a._buf is None
    a.fill(x)
    monkeypatch.setenv("MFT_STEM_FUSED_FILL", "0")
    b = Fn.StemCache(W, n, 84, DEV, pooled=True)
    assert not b.fused_fill
    b.fill(x)
    torch.cuda.synchronize()
    assert a.fused_fill                                                    # the launch was inside its domain
    assert torch.equal(a.pmax, b.pmax) and torch.equal(a.pmin, b.pmin)
    assert torch.allclose(a.mean, b.mean, rtol=0, atol=2e-6) and torch.allclose(a.m2, b.m2, rtol=2e-5, atol=1e-3)
    # float64 reference of the whole op on a few images
    k = min(n, 3)
    w64 = sd["trunk.0.weight"].double().to(DEV)
    c = torch.nn.functional.conv2d(x[:k].permute(0, 3, 1, 2).double(), w64, stride=2, padding=3)          # [k, 64, 42, 42]
    assert torch.allclose(a.mean[:k].double(), c.mean(dim=(2, 3)), atol=1e-5)
    assert torch.allclose(a.m2[:k].double(), ((c - c.mean(dim=(2, 3), keepdim=True)) ** 2).sum(dim=(2, 3)), rtol=1e-4, atol=1e-2)
    mx = torch.nn.functional.max_pool2d(c, 3, 2, 1).permute(0, 2, 3, 1)
    mn = -torch.nn.functional.max_pool2d(-c, 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.allclose(a.pmax[:k].double(), mx, atol=2e-5) and torch.allclose(a.pmin[:k].double(), mn, atol=2e-5)
    # a 224 x 224 cache is outside the fused kernel's domain and says so
    assert not Fn.StemCache(W, 2, 224, DEV, pooled=True).fused_fill


@pytest.mark.parametrize("rows,K,kv,cp,cv", [(7440, 160, 133, 192, 192), (480, 288, 266, 64, 48), (480, 480, 458, 32, 5), (7440, 96, 96, 32, 1),
                                             (300, 256, 229, 192, 0)])
def test_conv_wgrad_oihw_valid_corner(rows, K, kv, cp, cv):
    """mft_conv2d_wgrad_oihw(cin_valid, cout_valid): the GNN's linear layers carry zero-padded input features / output rows; the
    launch hands back the [cout_valid, cin_valid] corner contiguously -- equal, bit for bit, to slicing the full gradient
    (gnn.py:38,64-76: nn.Linear / 1x1 Conv2d .grad)."""
    x = rnd((rows, K), 301).to(DEV)
    x[:, kv:] = 0
    dy = rnd((rows, cp), 302).to(DEV)
    if cv:
        dy[:, cv:] = 0
    full = ops.conv2d_wgrad_oihw(x.view(rows, 1, 1, K), dy.view(rows, 1, 1, cp), cp, 1, 1, 1, 0).view(cp, K)
    got = ops.conv2d_wgrad_oihw(x.view(rows, 1, 1, K), dy.view(rows, 1, 1, cp), cp, 1, 1, 1, 0, cin_valid=kv, cout_valid=cv)
    assert got.shape == (cv or cp, kv, 1, 1) and got.is_contiguous()
    assert torch.equal(got.view(cv or cp, kv), full[:cv or cp, :kv])
    ref = dy.double().cpu().t() @ x.double().cpu()
    assert float((got.view(cv or cp, kv).double().cpu() - ref[:cv or cp, :kv]).abs().max()) < 2e-5 * float(ref.abs().max())


def test_conv_wgrad_multi_is_bit_identical_to_single_launches():
    """mft_conv2d_wgrad_oihw_multi (ops.WgradBatch: every weight gradient of a backward pass in one pair of launches per 16 jobs, the
    stem on its own) against the same problems launched one by one: bit for bit, for 3x3 / 1x1 / strided / stem / zero-padded linear
    jobs in one batch of 19 (two multi launches)."""
    jobs = [(105, 21, 64, 64, 3, 1, 1, 0, 0), (105, 21, 64, 128, 3, 2, 1, 0, 0), (105, 21, 64, 128, 1, 2, 0, 0, 0), (105, 6, 256, 512, 3, 2, 1, 0, 0),
            (105, 3, 512, 512, 3, 1, 1, 0, 0), (4, 84, 3, 64, 7, 2, 3, 0, 0), (7440, 1, 160, 192, 1, 1, 0, 133, 0), (7440, 1, 192, 192, 1, 1, 0, 0, 0),
            (7440, 1, 96, 32, 1, 1, 0, 0, 1), (480, 1, 288, 64, 1, 1, 0, 266, 48), (480, 1, 480, 32, 1, 1, 0, 458, 5), (105, 1, 512, 128, 1, 1, 0, 0, 0)]
    jobs = jobs + jobs[:7]
    wb = ops.WgradBatch(True)
    outs, refs = [], []
    for i, (n, H, Cin, Cout, k, stride, pad, cv, ov) in enumerate(jobs):
        x = nhwc(rnd((n, Cin, H, H), 400 + i)).to(DEV)
        OH = (H + 2 * pad - k) // stride + 1
        dy = nhwc(rnd((n, Cout, OH, OH), 500 + i)).to(DEV)
        if cv:
            x[..., cv:] = 0
        if ov:
            dy[..., ov:] = 0
        outs.append(wb.add(x, dy, Cout, k, k, stride, pad, cv, ov))
        refs.append(ops.conv2d_wgrad_oihw(x, dy, Cout, k, k, stride, pad, cv, ov))
    wb.flush()
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert o.shape == r.shape and torch.equal(o, r), (i, jobs[i])
    off = ops.WgradBatch(False)                            # disabled: add() launches at once
    x = nhwc(rnd((5, 64, 6, 6), 7)).to(DEV)
    dy = nhwc(rnd((5, 64, 6, 6), 8)).to(DEV)
    assert torch.equal(off.add(x, dy, 64, 3, 3, 1, 1), ops.conv2d_wgrad_oihw(x, dy, 64, 3, 3, 1, 1))


@pytest.mark.parametrize("rows,C,groups", [(945, 512, 1), (3780, 256, 1), (12705, 128, 1), (4 * 945, 512, 4), (46305, 64, 1)])
def test_bn_stats_and_backward_multi_are_bit_identical_to_single_launches(rows, C, groups):
    """mft_bn_stats_multi / mft_bn_backward_act_multi (SimpleBlock's BN2 + BNshortcut as one launch pair / triple in the meta-training
    step) against the same problems launched one by one: mean, rstd, running statistics and counter, dx, dgamma, dbeta bit for bit."""
    from meta_fine_tuning_amd import functional_bwd as FB
    xa, xb = rnd((rows, C), 601).to(DEV) * 1.5 + 0.3, rnd((rows, C), 602).to(DEV) * 0.7 - 0.2
    dy, ya = rnd((rows, C), 603).to(DEV), torch.relu(rnd((rows, C), 604)).to(DEV)
    ga, gb = (rnd((C,), 605).to(DEV) + 1.5), (rnd((C,), 606).to(DEV) + 1.5)

    def running():
        return torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
    ra, rb, ra2, rb2 = running(), running(), running(), running()
    (ma, sa), (mb, sb) = ops.bn_stats_multi([(xa, C, rows // groups, groups) + ra, (xb, C, rows // groups, groups) + rb])
    ma1, sa1 = ops.bn_stats(xa, C, rows // groups, groups, ra2[0], ra2[1], num_batches_tracked=ra2[2])
    mb1, sb1 = ops.bn_stats(xb, C, rows // groups, groups, rb2[0], rb2[1], num_batches_tracked=rb2[2])
    for got, ref in ((ma, ma1), (sa, sa1), (mb, mb1), (sb, sb1), (ra[0], ra2[0]), (ra[1], ra2[1]), (rb[0], rb2[0]), (rb[1], rb2[1])):
        assert torch.equal(got, ref)
    assert int(ra[2]) == int(rb[2]) == int(ra2[2]) == 1
    (dxa, dga, dba), (dxb, dgb, dbb) = ops.bn_backward_multi([(xa, dy, ya, C, rows, groups, ma, sa, ga, ops.ACT_RELU),
                                                              (xb, dy, ya, C, rows, groups, mb, sb, gb, ops.ACT_RELU)])
    ra_ = FB.bn_bwd(xa, dy, C, rows, ma, sa, ga, y_act=ya, act=ops.ACT_RELU, groups=groups)
    rb_ = FB.bn_bwd(xb, dy, C, rows, mb, sb, gb, y_act=ya, act=ops.ACT_RELU, groups=groups)
    for got, ref in zip((dxa, dga, dba, dxb, dgb, dbb), ra_ + rb_):
        assert torch.equal(got, ref)
