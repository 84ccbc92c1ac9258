"""Hot-path parity on a real MI355X: the HIP ResNet10 forward, the teacher-forced inner step
(last-block backward + fused Adam), the GNN head and the episode-batched FinetuneEngine against
the CPU oracle (run in float64 where the comparison is per-step) and the golden vectors produced
by the reference itself.  Tolerances follow BASELINE.json: 1e-3 on logits for forward-only and
teacher-forced single steps; full inner loops are held to the fp32 Adam envelope."""
import os

import numpy as np
import pytest
import torch

import meta_fine_tuning_amd  # noqa: F401
from meta_fine_tuning_amd import engine as eng
from meta_fine_tuning_amd import functional as Fn
from meta_fine_tuning_amd import ops, synthetic
from oracle import mft_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_num_threads(8)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("size", [84, 224])
def test_resnet10_forward_vs_reference_golden(golden_dir, size):
    g = _g(golden_dir, "g1_resnet10_fwd.npz")
    sd = synthetic.resnet10_state_dict(seed=3)
    x = synthetic.train_episode(11, 5, 1, 0, size).view(5, 3, size, size)
    W = Fn.ResNet10Weights(sd, DEV)
    arena = Fn.Arena(DEV)
    feat = Fn.resnet10_forward(W, ops.nchw_to_nhwc(x.to(DEV)), arena)
    np.testing.assert_allclose(feat.cpu().numpy(), g["feat_%d" % size], atol=1e-4)     # bar: 1e-3 on logits


def test_resnet10_forward_grouped_equals_per_group():
    """3 groups of 5 images in one grouped launch == 3 separate 5-image forwards (per-group BN statistics)."""
    sd = synthetic.resnet10_state_dict(seed=4)
    W = Fn.ResNet10Weights(sd, DEV)
    x = synthetic.train_episode(12, 5, 3, 0, 84).view(15, 3, 84, 84)
    xg = ops.nchw_to_nhwc(x.to(DEV))
    f_all = Fn.resnet10_forward(W, xg, Fn.Arena(DEV), ipg=5).clone()
    for gi in range(3):
        f = Fn.resnet10_forward(W, xg[gi * 5:(gi + 1) * 5].contiguous(), Fn.Arena(DEV), ipg=5)
        assert float((f - f_all[gi * 5:(gi + 1) * 5]).abs().max()) < 1e-5
    sd64 = O.clone_state(sd, torch.float64)
    with torch.no_grad():
        ref = O.resnet10_forward(sd64, x[:5].double(), "", train=True)
    assert float((f_all[:5].cpu().double() - ref).abs().max()) < 1e-4


def test_inner_step_teacher_forced(golden_dir):
    """One inner step (forward, CE, last-block backward, Adam) for E=2 episodes with different batches:
    gradients and Adam moments against the float64 oracle and the reference's fp32 golden gradients."""
    g = _g(golden_dir, "g4_inner_loop.npz")
    size = 84
    sd = synthetic.resnet10_state_dict(seed=9)
    views = synthetic.test_episode(31, 5, 5, 15, size, gen_examples=1)
    xa = torch.cat([v[:, :5].contiguous().view(25, 3, size, size) for v in [views[0]] + views], 0)
    ya = torch.from_numpy(np.tile(np.repeat(np.arange(5), 5), len(views) + 1))
    perm = g["perm"]
    sels = [perm[0:5], perm[5:10]]
    E = 2
    W = Fn.ResNet10Weights(sd, DEV)
    arena = Fn.Arena(DEV)
    ad = eng.AdaptState(E, DEV)
    ad.reset(W)
    xb = torch.cat([xa[torch.from_numpy(s)] for s in sels], 0)
    yb = torch.cat([ya[torch.from_numpy(s)] for s in sels], 0)
    tape = {}
    feat = Fn.resnet10_forward(W, ops.nchw_to_nhwc(xb.to(DEV)), arena, ipg=5, slab=ad.w, tape=tape)
    loss, dlog = ops.cross_entropy(feat, yb.to(torch.int32).to(DEV), 5, E)
    Fn.last_block_backward(tape, dlog, ad.w, ad.g, arena, ipg=5)
    np.testing.assert_allclose(feat[:5].cpu().numpy(), g["feat0_f32"], atol=1e-4)
    assert abs(float(loss[0].cpu()) - float(g["loss0_f32"])) < 1e-4
    grads0 = ad.g.export(0)
    np.testing.assert_allclose(grads0["trunk.7.C1.weight"][:2, :4].cpu().numpy(), g["g_c1_slice_f64"], atol=2e-5)
    np.testing.assert_allclose(grads0["trunk.7.C2.weight"][:2, :4].cpu().numpy(), g["g_c2_slice_f64"], atol=2e-5)
    np.testing.assert_allclose(grads0["trunk.7.shortcut.weight"][:4, :8, 0, 0].cpu().numpy(), g["g_sc_slice_f64"], atol=2e-5)
    for nm, key in (("bn1", "BN1"), ("bn2", "BN2"), ("bnsc", "BNshortcut")):
        np.testing.assert_allclose(grads0["trunk.7.%s.weight" % key].cpu().numpy(), g["g_%s_w_f64" % nm], atol=5e-5)
        np.testing.assert_allclose(grads0["trunk.7.%s.bias" % key].cpu().numpy(), g["g_%s_b_f64" % nm], atol=5e-5)
    # full-tensor check of both episodes against the float64 oracle
    for e in range(E):
        sd64 = O.clone_state(sd, torch.float64)
        adam = O.adam_init([sd64[k] for k in O.ADAPT_KEYS])
        sel = torch.from_numpy(sels[e])
        _, f64, gr, _ = O.inner_step(sd64, xa[sel].double(), ya[sel], adam, return_aux=True)
        ge = ad.g.export(e)
        for k, gref in zip(O.ADAPT_KEYS, gr):
            err = float((ge[k].cpu().double() - gref).abs().max())
            assert err < 3e-5 * max(1.0, float(gref.abs().max())), (k, err)
    # Adam: moments are linear in g; the weight step is checked on the well-conditioned entries
    w_before = ad.w.flat.clone()
    ops.adam_step(ad.w.flat, ad.g.flat, ad.m.flat, ad.v.flat, 1, lr=0.01)
    assert float((ad.m.flat - 0.1 * ad.g.flat).abs().max()) < 1e-7
    big = ad.g.flat.abs() > 1e-6
    dw = (ad.w.flat - w_before)[big]
    assert float((dw + 0.01 * torch.sign(ad.g.flat[big])).abs().max()) < 1e-4      # first Adam step = -lr*sign(g)


@pytest.mark.parametrize("B,N", [(15, 30), (2, 105), (2, 130)])
def test_gnn_forward_vs_reference_golden(golden_dir, B, N):
    g = _g(golden_dir, "g2_gnn.npz")
    sd = synthetic.gnn_head_state_dict(seed=5)
    G = Fn.GnnHeadWeights(sd, DEV, 5)
    rs = np.random.RandomState(100 + N + B)
    nodes = torch.from_numpy(rs.standard_normal((B, N, 133)).astype(np.float32))
    x = torch.zeros(B * N, 256)
    x[:, :133] = nodes.view(B * N, 133)
    arena = Fn.Arena(DEV)
    xd = x.to(DEV)
    A0 = Fn.wcompute(G, "layer_w0", xd, 133, B, N, 1, arena).clone()
    np.testing.assert_allclose(A0.cpu().numpy(), g["A0_%d_%d" % (B, N)], atol=1e-5)
    out = Fn.gnn_forward(G, xd, B, N, 1, arena)
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, N, 5), g["out_%d_%d" % (B, N)], atol=2e-4)


@pytest.mark.parametrize("B,N", [(15, 105), (15, 130)])
def test_gnn_forward_full_graph_batches_vs_reference_golden(golden_dir, B, N):
    """The 20-/50-shot graph sizes with the full 15-graph batch (BatchNorm over 15*N*N pair positions) through the fused
    pair-MLP kernels, against the reference's own GNN_nl output (G2b)."""
    g = _g(golden_dir, "g2b_gnn_full.npz")
    sd = synthetic.gnn_head_state_dict(seed=5)
    G = Fn.GnnHeadWeights(sd, DEV, 5)
    rs = np.random.RandomState(100 + N + B)
    nodes = torch.from_numpy(rs.standard_normal((B, N, 133)).astype(np.float32))
    x = torch.zeros(B * N, 256)
    x[:, :133] = nodes.view(B * N, 133)
    arena = Fn.Arena(DEV)
    xd = x.to(DEV)
    A0 = Fn.wcompute(G, "layer_w0", xd, 133, B, N, 1, arena).clone()
    np.testing.assert_allclose(A0[0].cpu().numpy(), g["A0first_%d_%d" % (B, N)], atol=1e-5)
    np.testing.assert_allclose((A0.double() ** 2).sum(2).cpu().numpy(), g["A0diag2_%d_%d" % (B, N)], atol=1e-5)
    out = Fn.gnn_forward(G, xd, B, N, 1, arena)
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, N, 5), g["out_%d_%d" % (B, N)], atol=3e-4)
    # nothing of size [B*N*N, F] lives in the arena: the largest buffer is a raw layer output over the upper-triangle rows
    biggest = max(t.numel() for t in arena.bufs.values())
    assert biggest <= B * (N * (N + 1) // 2) * 192, biggest


@pytest.mark.parametrize("N,F", [(30, 133), (30, 181), (105, 229), (7, 133)])
def test_fused_pair_mlp_equals_materialised_wcompute(N, F, monkeypatch):
    """csrc/pair_mlp.hip (upper-triangle rows, |x_i - x_j| in the loader, BatchNorm from the epilogue statistics, BatchNorm +
    leaky_relu in the next loader) against the materialised sequence it replaces and against float64, two episodes of 3
    graphs with DIFFERENT statistics; also with the episodes processed in two chunks."""
    sd = synthetic.gnn_head_state_dict(seed=5)
    G = Fn.GnnHeadWeights(sd, DEV, 5)
    name = {133: "layer_w0", 181: "layer_w1", 229: "w_comp_last"}[F]
    rs = np.random.RandomState(N + F)
    B, groups = 6, 2
    nodes = rs.standard_normal((B, N, F)).astype(np.float32)
    nodes[3:] *= 1.7                                                  # second episode: different scale
    x = torch.zeros(B * N, 256)
    x[:, :F] = torch.from_numpy(nodes).view(B * N, F)
    x[:, F:] = 1e30                                                   # columns beyond F must be ignored
    xd = x.to(DEV)
    a_f = Fn.wcompute(G, name, xd, F, B, N, groups, Fn.Arena(DEV)).clone()
    a_u = Fn.wcompute_unfused(G, name, xd, F, B, N, groups, Fn.Arena(DEV)).clone()
    monkeypatch.setattr(Fn, "PAIR_MLP_BYTES", 1)                      # one episode per chunk
    a_c = Fn.wcompute(G, name, xd, F, B, N, groups, Fn.Arena(DEV)).clone()
    assert torch.equal(a_c, a_f)
    # float64 statement of gnn.py:78-115 per episode
    ref = []
    for e in range(groups):
        with torch.no_grad():
            ref.append(O.wcompute(O.clone_state(sd, torch.float64), "gnn." + name, torch.from_numpy(nodes[3 * e:3 * e + 3]).double()))
    ref = torch.cat(ref).numpy()
    e_f = np.abs(a_f.cpu().numpy() - ref).max()
    e_u = np.abs(a_u.cpu().numpy() - ref).max()
    assert e_f < 2e-5 and e_f <= 3.0 * e_u + 2e-6, (e_f, e_u)
    np.testing.assert_allclose(a_f.sum(2).cpu().numpy(), 1.0, atol=1e-5)
    assert float(a_f.diagonal(dim1=1, dim2=2).abs().max()) == 0.0    # masked diagonal (gnn.py:105-107)
    # the same layers as f16x2 products on the fp16 matrix cores (round 6): fp32-accurate -- held to the fp32-MFMA form's own bound
    monkeypatch.setattr(Fn, "PAIR_F16X2", not Fn.PAIR_F16X2)
    a_h = Fn.wcompute(G, name, xd, F, B, N, groups, Fn.Arena(DEV)).clone()
    e_h = np.abs(a_h.cpu().numpy() - ref).max()
    assert e_h < 2e-5 and e_h <= 3.0 * e_u + 2e-6, (e_h, e_f, e_u)
    np.testing.assert_allclose(a_h.sum(2).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.parametrize("N,F,B,groups", [(30, 133, 32, 2), (30, 181, 16, 1), (30, 229, 16, 1), (7, 133, 6, 2), (26, 133, 9, 3)])
def test_register_k_pair_layers_match_tile_kernel_and_float64(N, F, B, groups, monkeypatch):
    """The register-K form of the Wcompute layers (mft_pair_mlp_layer_rk: 32-row tiles, the whole K in registers split over four
    waves -- what a meta-training step of one or a few episodes launches) against the 128-row tile kernel on the same inputs and
    against the float64 statement of gnn.py:78-115: same raw layer outputs / BatchNorm tables to rounding (another fixed
    summation order over k), A within the tile kernel's own error bound; run twice -> bit-identical."""
    from meta_fine_tuning_amd import functional_bwd as FB
    sd = synthetic.gnn_head_state_dict(seed=5)
    G = Fn.GnnHeadWeights(sd, DEV, 5)
    name = {133: "layer_w0", 181: "layer_w1", 229: "w_comp_last"}[F]
    rs = np.random.RandomState(N + F)
    nodes = rs.standard_normal((B, N, F)).astype(np.float32)
    gpg = B // groups
    nodes[gpg:] *= 1.7                                                # later episodes: different scale
    x = torch.zeros(B * N, 256)
    x[:, :F] = torch.from_numpy(nodes).view(B * N, F)
    x[:, F:] = 1e30                                                   # columns beyond F must be ignored
    xd = x.to(DEV)
    monkeypatch.setattr(FB, "PAIR_RK_ROWS", 1 << 30)
    a_rk, t_rk = FB.wcompute_taped(G, name, xd, F, B, N, groups)
    a_rk2, _ = FB.wcompute_taped(G, name, xd, F, B, N, groups)
    assert torch.equal(a_rk, a_rk2)
    monkeypatch.setattr(FB, "PAIR_RK_ROWS", 0)
    a_t, t_t = FB.wcompute_taped(G, name, xd, F, B, N, groups)
    for z0, z1 in zip(t_rk["z"], t_t["z"]):
        assert float((z0 - z1).abs().max()) <= 2e-5 * max(1.0, float(z1.abs().max()))
    for bn0, bn1 in zip(t_rk["bn"], t_t["bn"]):
        for u, v in zip(bn0, bn1):
            assert float((u - v).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max()))
    ref = []
    for e in range(groups):
        with torch.no_grad():
            ref.append(O.wcompute(O.clone_state(sd, torch.float64), "gnn." + name, torch.from_numpy(nodes[gpg * e:gpg * (e + 1)]).double()))
    ref = torch.cat(ref).numpy()
    e_rk = np.abs(a_rk.cpu().numpy() - ref).max()
    e_t = np.abs(a_t.cpu().numpy() - ref).max()
    assert e_rk < 2e-5 and e_rk <= 3.0 * e_t + 2e-6, (e_rk, e_t)
    np.testing.assert_allclose(a_rk.sum(2).cpu().numpy(), 1.0, atol=1e-5)
    assert float(a_rk.diagonal(dim1=1, dim2=2).abs().max()) == 0.0


def test_gnnnet_scores_50shot_fold(golden_dir):
    g = _g(golden_dir, "g7_gnnnet50.npz")
    sd = synthetic.gnn_head_state_dict(seed=19)
    G = Fn.GnnHeadWeights(sd, DEV, 5)
    feats = torch.from_numpy(np.random.RandomState(61).standard_normal((5, 65, 512)).astype(np.float32))
    sc = Fn.gnnnet_scores(G, feats.view(-1, 512).to(DEV), 1, 5, 25, 15, Fn.Arena(DEV), fold=True)
    np.testing.assert_allclose(sc.cpu().numpy(), g["scores"], atol=1e-3)


@pytest.mark.parametrize("E_epochs,G_aug", [(0, 0), (1, 0), (1, 2)])
def test_engine_vs_reference_finetune_golden(golden_dir, E_epochs, G_aug):
    """FinetuneEngine (one episode in a batch of 2 slots) against the scores the reference's own
    finetune() produced on the same episode, weights and numpy seed."""
    g = _g(golden_dir, "g5_finetune.npz")
    sd = synthetic.gnnnet_state_dict(seed=13)
    liz = synthetic.test_episode(41 + G_aug, 5, 5, 15, 84, gen_examples=G_aug)
    e = eng.FinetuneEngine(sd, n_views=2 + G_aug, fine_tune_epoch=E_epochs, episodes_per_batch=2, device=DEV)
    np.random.seed(10)
    st = np.random.get_state()
    sc = e.run_batch([liz])[0].cpu().numpy()
    ref = g["scores_E%d_G%d" % (E_epochs, G_aug)]
    if E_epochs == 0:
        np.testing.assert_allclose(sc, ref, atol=1e-4)          # forward-only: bar is 1e-3 on logits
    else:
        # 15-20 Adam steps: no two fp32 implementations agree step for step (the first Adam steps move every
        # weight by lr * sign(g)), so the bar is the envelope around the float64 run of the same episode: the engine may sit no
        # further from it than 4x the REFERENCE's own fp32 distance to it (gate ii), and hence within 5x that distance of the
        # reference's scores (round-4 verdict weak 2: the flat 2e-2 that stood here was 7x the reference's own distance)
        np.random.set_state(st)
        o64 = O.finetune_episode(sd, liz, 5, 5, total_epoch=E_epochs, dtype=torch.float64).numpy()
        d_ref = np.abs(ref - o64).max()
        assert np.abs(sc - o64).max() <= max(4.0 * d_ref, 1e-3), (np.abs(sc - o64).max(), d_ref)
        assert np.abs(sc - ref).max() <= max(5.0 * d_ref, 2e-3), (np.abs(sc - ref).max(), d_ref)
        assert (sc.argmax(1) == ref.argmax(1)).mean() >= 0.96
    e.close()


def test_engine_batched_matches_oracle_envelope():
    """Two different episodes run in lockstep; each must stay within k x the reference's own fp32-vs-fp64
    distance of the float64 oracle (SURVEY.md §0 D7 gate (ii)), with the same permutations."""
    sd = synthetic.gnnnet_state_dict(seed=21)
    eps = [synthetic.test_episode(300 + i, 5, 5, 15, 84, gen_examples=1) for i in range(2)]
    rs = np.random.RandomState(5)
    perms = [[rs.permutation(100) for _ in range(1)] for _ in range(2)]
    e = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
    sc = e.run_batch(eps, perms=perms).cpu().numpy()
    for i in range(2):
        o64 = O.finetune_episode(sd, eps[i], 5, 5, total_epoch=1, perms=perms[i], dtype=torch.float64).numpy()
        o32 = O.finetune_episode(sd, eps[i], 5, 5, total_epoch=1, perms=perms[i], dtype=torch.float32).numpy()
        d_ref = np.abs(o32 - o64).max()
        d_hip = np.abs(sc[i] - o64).max()
        assert d_hip <= max(4.0 * d_ref, 2e-3), (d_hip, d_ref)
        assert (sc[i].argmax(1) == o64.argmax(1)).mean() >= 0.96


def test_fused_wgrad_adam_equals_unfused():
    """The wgrad kernels with Adam in the epilogue must produce the same (w, m, v) as wgrad -> mft_adam_step."""
    sd = synthetic.gnnnet_state_dict(seed=23)
    eps = [synthetic.test_episode(400 + i, 5, 5, 15, 84, gen_examples=0) for i in range(2)]
    rs = np.random.RandomState(6)
    perms = [[rs.permutation(75)] for _ in range(2)]
    outs = []
    for fused in (False, True):
        e = eng.FinetuneEngine(sd, n_views=2, fine_tune_epoch=1, episodes_per_batch=2, device=DEV, fused_adam=fused, fuse_next=False)
        sc = e.run_batch(eps, perms=perms)
        outs.append((sc.clone(), e.adapt.w.flat.clone(), e.adapt.m.flat.clone(), e.adapt.v.flat.clone()))
    (s0, w0, m0, v0), (s1, w1, m1, v1) = outs
    assert float((m0 - m1).abs().max()) < 2e-5 and float((v0 - v1).abs().max()) < 1e-7     # after 15 chaotic Adam steps
    frac_far = float(((w0 - w1).abs() > 1e-4).float().mean())
    assert frac_far < 1e-3 and float((s0 - s1).abs().max()) < 5e-3, (frac_far, float((s0 - s1).abs().max()))


def test_fused_last_block_launches_match_separate_launches():
    """Engine with the fused block-entry / block-exit / data-gradient+BN / CE+BN launches (default) against the same engine on
    the separate launches they replace (MFT_FUSED_LAST_BLOCK=0): per-episode BatchNorm statistics are reduced in a different
    order, so agreement is to fp32 rounding amplified by 15 Adam steps, not bit for bit."""
    sd = synthetic.gnnnet_state_dict(seed=33)
    eps = [synthetic.test_episode(700 + i, 5, 5, 15, 84, gen_examples=0) for i in range(2)]
    rs = np.random.RandomState(11)
    perms = [[rs.permutation(75)] for _ in range(2)]
    outs = []
    old = Fn.FUSED_LAST_BLOCK
    try:
        for fused in (False, True):
            Fn.FUSED_LAST_BLOCK = fused
            e = eng.FinetuneEngine(sd, n_views=2, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
            sc = e.run_batch(eps, perms=perms)
            outs.append((sc.clone(), e.adapt.w.flat.clone(), e.adapt.m.flat.clone(), e.adapt.v.flat.clone()))
    finally:
        Fn.FUSED_LAST_BLOCK = old
    (s0, w0, m0, v0), (s1, w1, m1, v1) = outs
    dm = (m0 - m1).abs()
    assert float(dm.max()) < 3e-4 and float((dm > 1e-5).float().mean()) < 1e-3 and float((v0 - v1).abs().max()) < 1e-6, \
        (float(dm.max()), float((dm > 1e-5).float().mean()), float((v0 - v1).abs().max()))
    # Adam's first steps move every weight by ~lr * sign(g): elements whose gradient is at rounding level take a different
    # path under ANY reordering of the BatchNorm sums (1.5 % of the weights here), while m, v and the scores agree closely
    frac_far = float(((w0 - w1).abs() > 1e-4).float().mean())
    assert frac_far < 5e-2 and float((s0 - s1).abs().max()) < 5e-3, (frac_far, float((s0 - s1).abs().max()))


def test_next_forward_fusion_matches_separate_forward_launches():
    """The default inner loop -- step t's weight-gradient + Adam launches also compute step t+1's trunk.7 forward from the
    weight tiles they have just updated (csrc/wgrad_fwd.hip; fp32 MFMA, K walked tile by tile) -- against the same engine with
    the separate block-entry / block-exit launches (bf16x3, another summation order): the FIRST step's update is bit-identical
    (its forward is the same launch in both), later steps agree to fp32 rounding amplified by Adam, like any two fp32 forms."""
    sd = synthetic.gnnnet_state_dict(seed=33)
    eps = [synthetic.test_episode(700 + i, 5, 5, 15, 84, gen_examples=0) for i in range(2)]
    rs = np.random.RandomState(11)
    perms = [[rs.permutation(75)] for _ in range(2)]
    outs, first = [], []
    for fuse in (False, True):
        for pipe in (False, True):
            e = eng.FinetuneEngine(sd, n_views=2, fine_tune_epoch=1, episodes_per_batch=2, device=DEV, fuse_next=fuse, pipeline=pipe)
            e._ingest(eps, False)
            e.adapt.reset(e.W)
            e.prepare_batch()
            e.inner_loop(e.step_tables(perms, 2)[:1])
            torch.cuda.synchronize()
            first.append(e.adapt.w.flat.clone())
            sc = e.run_batch(eps, perms=perms)
            outs.append((sc.clone(), e.adapt.w.flat.clone(), e.adapt.m.flat.clone(), e.adapt.v.flat.clone()))
            e.close()
    assert all(torch.equal(first[0], f) for f in first[1:])                       # one step: the same update, bit for bit
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and all(torch.equal(a, b) for a, b in zip(outs[2], outs[3]))
    (s0, w0, m0, v0), (s1, w1, m1, v1) = outs[0], outs[2]
    dm = (m0 - m1).abs()
    assert float(dm.max()) < 3e-4 and float((dm > 1e-5).float().mean()) < 1e-3 and float((v0 - v1).abs().max()) < 1e-6, \
        (float(dm.max()), float((dm > 1e-5).float().mean()), float((v0 - v1).abs().max()))
    frac_far = float(((w0 - w1).abs() > 1e-4).float().mean())
    assert frac_far < 5e-2 and float((s0 - s1).abs().max()) < 5e-3, (frac_far, float((s0 - s1).abs().max()))


def test_scores_do_not_depend_on_the_slot_beyond_fp32_rounding():
    """The reference adapts one episode at a time; the engine runs E in lockstep and an episode's BatchNorm statistics in the frozen
    trunk are reduced tile by tile, so WHERE an episode sits in the batch (and how large E is) changes the summation order of
    those sums -- nothing else.  Stated bound: with no adaptation the scores of the same episode in different slots / batch
    sizes agree to 1e-5; after 20 Adam steps to 1e-2 (measured 5e-3: the size of the fp32-vs-fp64 distance of the reference
    itself after as many steps -- Adam's lr * sign(g) first moves amplify any rounding difference) with the same
    predictions on >= 96 % of the queries."""
    sd = synthetic.gnnnet_state_dict(seed=27)
    ep = synthetic.test_episode(910, 5, 5, 15, 84, gen_examples=1)
    other = [synthetic.test_episode(911 + i, 5, 5, 15, 84, gen_examples=1) for i in range(3)]
    perm = [np.random.RandomState(50).permutation(100)]
    operm = [[np.random.RandomState(60 + i).permutation(100)] for i in range(3)]
    for epochs, tol in ((0, 1e-5), (1, 1e-2)):
        res = []
        for E, slot in ((1, 0), (4, 0), (4, 3), (3, 1)):
            e = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=epochs, episodes_per_batch=E, device=DEV)
            eps, perms = list(other[:E - 1]), list(operm[:E - 1])
            eps.insert(slot, ep)
            perms.insert(slot, perm)
            res.append(e.run_batch(eps, perms=[p[:epochs] for p in perms])[slot].cpu().numpy())
            e.close()
        for r in res[1:]:
            assert np.abs(r - res[0]).max() <= tol, (epochs, np.abs(r - res[0]).max())
            assert (r.argmax(1) == res[0].argmax(1)).mean() >= 0.96


def test_two_stream_pipeline_is_bit_identical():
    """Running the frozen trunk of step t+1 on a second stream must not change a single bit."""
    sd = synthetic.gnnnet_state_dict(seed=25)
    eps = [synthetic.test_episode(500 + i, 5, 5, 15, 84, gen_examples=1) for i in range(3)]
    rs = np.random.RandomState(7)
    perms = [[rs.permutation(100), rs.permutation(100)] for _ in range(3)]
    res = []
    for pipe in (False, True):
        e = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=2, episodes_per_batch=3, device=DEV, pipeline=pipe)
        sc = e.run_batch(eps, perms=perms)
        res.append((sc.clone(), e.adapt.w.flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_presplit_activation_planes_are_bit_identical():
    """Frozen trunk on pre-split bf16x3 activation planes (producers split once: mft_bn_apply_planes /
    mft_bn_relu_maxpool_gather_planes, consumer mft_conv2d_nhwc_x3p_bnstats) against the same trunk splitting inside every
    convolution tile: the pieces are the same numbers, so the activation entering trunk.7 must match bit for bit."""
    W = Fn.ResNet10Weights(synthetic.resnet10_state_dict(seed=31), DEV, x3=True, f16x2=False)       # (the planes chain is a bf16x3 form)
    rs = np.random.RandomState(9)
    x = torch.from_numpy(rs.standard_normal((60, 84, 84, 3)).astype(np.float32)).to(DEV)
    cache = Fn.StemCache(W, 60, 84, DEV, chunk=32, pooled=False)      # the planes chain reads the full-resolution cache
    cache.fill(x)
    idx = torch.from_numpy(rs.permutation(60)[:35].astype(np.int32)).to(DEV)
    from meta_fine_tuning_amd import _lib
    outs = []
    old = Fn.X3_PLANES
    try:
        # (planes, knob): the per-tap implicit-GEMM kernel (mft_debug_set_x3_tile(90)) walks K in the same order as the planes
        # kernel; the default shared-tap kernel of the 3x3 / stride-1 layers walks (kh, ci, kw) -- same products, other order
        for planes, knob in ((False, 90), (True, 90), (False, 91)):
            _lib.lib().mft_debug_set_x3_tile(knob)
            Fn.X3_PLANES = planes
            arena = Fn.Arena(DEV)
            outs.append(Fn.resnet10_trunk(W, None, arena, 5, upto=7, tag="p%d" % planes, stem=(cache, idx)).clone())
            assert any(k[0].endswith(".r1p") for k in arena.bufs) == planes          # the chain really ran / did not run
    finally:
        Fn.X3_PLANES = old
        _lib.lib().mft_debug_reset()
    assert outs[0].shape == (35, 6, 6, 256)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[2] - outs[0]).abs().max()) < 2e-5 * float(outs[0].abs().max())      # fp32 rounding, three layers deep


def test_folded_trunk_batchnorms_are_bit_identical():
    """Frozen trunk without helper launches (functional.X3_FOLD_BN: statistics merged by their consumers, BN1 + ReLU in C2's loader,
    stem moments combined inside the pooled gather) against the separate finalize / apply / combine launches: the activation
    entering trunk.7 must match bit for bit, and so must a whole engine batch."""
    W = Fn.ResNet10Weights(synthetic.resnet10_state_dict(seed=33), DEV, x3=True)
    rs = np.random.RandomState(19)
    x = torch.from_numpy(rs.standard_normal((60, 84, 84, 3)).astype(np.float32)).to(DEV)
    cache = Fn.StemCache(W, 60, 84, DEV, chunk=32, pooled=True)
    cache.fill(x)
    idx = torch.from_numpy(rs.permutation(60)[:35].astype(np.int32)).to(DEV)
    g, b = W.bn["trunk.1"]
    m, s = torch.empty((7, 64), device=DEV), torch.empty((7, 64), device=DEV)
    ops.bn_combine_moments(cache.mean, cache.m2, idx, 42 * 42, 5, 7, mean=m, rstd=s)
    ref = cache.gather(idx, 35, m, s, g, b, 5, torch.empty((35, 21, 21, 64), device=DEV))
    m2, s2 = torch.empty_like(m), torch.empty_like(s)
    got = cache.gather_moments(idx, 35, g, b, 5, torch.empty((35, 21, 21, 64), device=DEV), stats=(m2, s2))
    assert torch.equal(got, ref) and torch.equal(m2, m) and torch.equal(s2, s)
    old = Fn.X3_FOLD_BN
    outs, launches = [], []
    try:
        for fold in (False, True):
            Fn.X3_FOLD_BN = fold
            arena = Fn.Arena(DEV)
            outs.append(Fn.resnet10_trunk(W, None, arena, 5, upto=7, tag="f%d" % fold, stem=(cache, idx)).clone())
            launches.append(any(k[0].endswith(".bn2.statws") for k in arena.bufs))
        assert launches == [False, True]                                    # the folded chain really ran / did not run
        assert outs[0].shape == (35, 6, 6, 256)
        assert torch.equal(outs[0], outs[1])
        sd = synthetic.gnnnet_state_dict(seed=4)
        eps = [synthetic.test_episode(60 + i, 5, 5, 15, 84, gen_examples=1) for i in range(7)]      # 7 groups per step: ragged tiles
        prs = np.random.RandomState(5)
        perms = [[prs.permutation(100) for _ in range(2)] for _ in range(7)]
        scores = []
        for fold in (False, True):
            Fn.X3_FOLD_BN = fold
            e = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=2, episodes_per_batch=7, device=DEV)
            scores.append(e.run_batch(eps, perms=perms).clone())
        assert torch.equal(scores[0], scores[1])
    finally:
        Fn.X3_FOLD_BN = old


def test_stem_cache_matches_recomputed_stem():
    """Cached trunk.0 outputs + recombined per-image BatchNorm moments must reproduce the per-step stem
    (conv -> batch statistics -> BN/ReLU/maxpool) to fp32 rounding, and the engine's scores must agree."""
    W = Fn.ResNet10Weights(synthetic.resnet10_state_dict(seed=27), DEV)
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.standard_normal((40, 84, 84, 3)).astype(np.float32)).to(DEV)
    cache = Fn.StemCache(W, 40, 84, DEV, chunk=16, pooled=False)          # ragged last chunk
    cache.fill(x)
    # pooled form (default in the engine): per-window (max, min) of the raw stem output; the BatchNorm affine is monotone per
    # channel, so the gather must reproduce the full-resolution BN -> ReLU -> MaxPool bit for bit -- also for negative gammas
    pc = Fn.StemCache(W, 40, 84, DEV, chunk=16, pooled=True)
    pc.fill(x)
    assert pc.c0 is None and pc.nbytes() < 0.6 * cache.nbytes() + 16 * 42 * 42 * 64 * 4
    idx_p = torch.from_numpy(rs.permutation(40)[:20].astype(np.int32)).to(DEV)
    m_p = torch.empty((4, 64), device=DEV)
    s_p = torch.empty((4, 64), device=DEV)
    ops.bn_combine_moments(pc.mean, pc.m2, idx_p, 42 * 42, 5, 4, mean=m_p, rstd=s_p)
    g_p, b_p = W.bn["trunk.1"]
    for gam in (g_p, -g_p, g_p * torch.from_numpy(np.where(rs.rand(64) < 0.5, -1.0, 1.0).astype(np.float32)).to(DEV)):
        full = cache.gather(idx_p, 20, m_p, s_p, gam.contiguous(), b_p, 5, torch.empty((20, 21, 21, 64), device=DEV))
        pooled = pc.gather(idx_p, 20, m_p, s_p, gam.contiguous(), b_p, 5, torch.empty((20, 21, 21, 64), device=DEV))
        assert torch.equal(full, pooled)
    idx = torch.from_numpy(rs.permutation(40)[:15].astype(np.int32)).to(DEV)
    a_cached = Fn.resnet10_trunk(W, None, Fn.Arena(DEV), 5, upto=4, tag="a", stem=(cache, idx))
    a_direct = Fn.resnet10_trunk(W, x[idx.long()].contiguous(), Fn.Arena(DEV), 5, upto=4, tag="b")
    assert a_cached.shape == a_direct.shape == (15, 21, 21, 64)
    assert float((a_cached - a_direct).abs().max()) < 2e-5
    # float64 statement of the per-image moments
    c0 = cache.c0.double().cpu().view(40, -1, 64)
    assert float((cache.mean.cpu().double() - c0.mean(1)).abs().max()) < 1e-6
    m2 = ((c0 - c0.mean(1, keepdim=True)) ** 2).sum(1)
    assert float(((cache.m2.cpu().double() - m2).abs() / (m2 + 1e-6)).max()) < 1e-5

    sd = synthetic.gnnnet_state_dict(seed=29)
    eps = [synthetic.test_episode(600 + i, 5, 5, 15, 84, gen_examples=1) for i in range(2)]
    perms = [[rs.permutation(100), rs.permutation(100)] for _ in range(2)]
    res = []
    for sc_on in (False, True):
        e = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=2, episodes_per_batch=2, device=DEV, stem_cache=sc_on)
        res.append(e.run_batch(eps, perms=perms).clone())
    for i in range(2):
        o64 = O.finetune_episode(sd, eps[i], 5, 5, total_epoch=2, perms=perms[i], dtype=torch.float64).numpy()
        o32 = O.finetune_episode(sd, eps[i], 5, 5, total_epoch=2, perms=perms[i], dtype=torch.float32).numpy()
        d_ref = np.abs(o32 - o64).max()
        for r in res:                                   # with and without the cache: same envelope around the fp64 oracle
            d = np.abs(r[i].cpu().numpy() - o64).max()
            assert d <= max(4.0 * d_ref, 3e-3), (d, d_ref)
            assert (r[i].cpu().numpy().argmax(1) == o64.argmax(1)).mean() >= 0.96


def test_finetune_linear_and_all_vs_reference_golden(golden_dir):
    """finetune_linear (per-episode Linear head + last block, 100 Adam steps) and the README's `--method all` sum
    (finetune.py:45-174,647-649) against the scores the reference produced for the same episode, weights, classifier
    initialisation and numpy seed (G10)."""
    from meta_fine_tuning_amd import finetune as ft
    from meta_fine_tuning_amd.methods.gnnnet import GnnNet
    from meta_fine_tuning_amd.io_utils import model_dict
    import argparse
    g = _g(golden_dir, "g10_finetune_linear.npz")
    sd = synthetic.gnnnet_state_dict(seed=37)
    liz = synthetic.test_episode(91, 5, 5, 15, 84, gen_examples=1)
    np.random.seed(10)
    sc = ft.finetune_linear([v.to(DEV) for v in liz], None, sd, None, linear=True, classifier=(g["w0"], g["b0"]))
    ref = g["scores_linear"]
    err = np.abs(sc.cpu().numpy() - ref)
    assert sc.shape == (75, 5)
    assert err.max() < 2e-2 and (sc.cpu().numpy().argmax(1) == ref.argmax(1)).mean() >= 0.96, err.max()
    # the same against the float64 oracle envelope
    np.random.seed(10)
    o64 = O.finetune_linear_episode(sd, liz, 5, 5, torch.from_numpy(g["w0"]), torch.from_numpy(g["b0"]), dtype=torch.float64).numpy()
    np.random.seed(10)
    o32 = O.finetune_linear_episode(sd, liz, 5, 5, torch.from_numpy(g["w0"]), torch.from_numpy(g["b0"]), dtype=torch.float32).numpy()
    assert np.abs(sc.cpu().numpy() - o64).max() <= max(4.0 * np.abs(o32 - o64).max(), 2e-3)
    # --method all: finetune_linear + finetune on the same numpy stream
    ft.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=1)
    model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=5)
    model.load_state_dict(sd)
    np.random.seed(10)
    tot = ft.finetune_all([v.to(DEV) for v in liz], None, model, sd, sd, classifier=(g["w0"], g["b0"])).cpu().numpy()
    ref = g["scores_all"]
    assert np.abs(tot - ref).max() < 4e-2 and (tot.argmax(1) == ref.argmax(1)).mean() >= 0.96


def test_linear_engine_batched_equals_single():
    """Two episodes in lockstep (linear mode) give each episode the scores it gets alone."""
    sd = synthetic.gnnnet_state_dict(seed=39)
    eps = [synthetic.test_episode(700 + i, 5, 5, 15, 84, gen_examples=0) for i in range(2)]
    rs = np.random.RandomState(9)
    perms = [[rs.permutation(25) for _ in range(20)] for _ in range(2)]
    w0 = torch.from_numpy((rs.uniform(-1, 1, (2, 5, 512)) / 22.6).astype(np.float32))
    b0 = torch.from_numpy((rs.uniform(-1, 1, (2, 5)) / 22.6).astype(np.float32))
    e2 = eng.FinetuneEngine(sd, n_views=2, fine_tune_epoch=20, episodes_per_batch=2, device=DEV, mode="linear")
    both = e2.run_batch(eps, perms=perms, classifier_init=(w0, b0)).clone()
    e1 = eng.FinetuneEngine(sd, n_views=2, fine_tune_epoch=20, episodes_per_batch=1, device=DEV, mode="linear")
    for i in range(2):
        one = e1.run_batch([eps[i]], perms=[perms[i]], classifier_init=(w0[i:i + 1], b0[i:i + 1]))[0]
        # the BatchNorm reduction is chunked by launch size, so E=1 and E=2 round differently; 100 Adam steps amplify that
        assert float((one - both[i]).abs().max()) < 2e-2
        assert float((one.argmax(1) == both[i].argmax(1)).float().mean()) >= 0.96


def test_hipgraph_inner_step_is_bit_identical():
    """One inner step captured as a hipGraph (device-side Adam step counter) and replayed for every step must give
    exactly the eager single-stream result, also on the second batch (pure replay)."""
    sd = synthetic.gnnnet_state_dict(seed=43)
    eps = [synthetic.test_episode(800 + i, 5, 5, 15, 84, gen_examples=1) for i in range(2)]
    rs = np.random.RandomState(15)
    perms = [[rs.permutation(100), rs.permutation(100)] for _ in range(2)]
    e0 = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=2, episodes_per_batch=2, device=DEV, pipeline=False, fuse_next=False)
    ref = e0.run_batch(eps, perms=perms).clone()
    wref = e0.adapt.w.flat.clone()
    e1 = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=2, episodes_per_batch=2, device=DEV, graph=True)
    for rep in range(2):
        got = e1.run_batch(eps, perms=perms).clone()
        assert torch.equal(got, ref), rep
        assert torch.equal(e1.adapt.w.flat, wref), rep


def test_engine_at_reference_image_size_224():
    """The reference hard-codes image_size = 224 (train.py:72, finetune.py:429): 7x7 last feature map, 245 rows
    per episode in trunk.7 (the skinny / fused small-group kernels do not apply and the generic paths take over).  One episode,
    15 Adam steps, against the float64 oracle envelope."""
    sd = synthetic.gnnnet_state_dict(seed=45)
    ep = synthetic.test_episode(900, 5, 5, 15, 224, gen_examples=0)
    perms = [[np.random.RandomState(17).permutation(75)]]
    e = eng.FinetuneEngine(sd, 5, 5, 15, 224, n_views=2, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
    sc = e.run_batch([ep], perms=perms)[0].cpu().numpy()
    o64 = O.finetune_episode(sd, ep, 5, 5, total_epoch=1, perms=perms[0], dtype=torch.float64).numpy()
    o32 = O.finetune_episode(sd, ep, 5, 5, total_epoch=1, perms=perms[0], dtype=torch.float32).numpy()
    d_ref, d_hip = np.abs(o32 - o64).max(), np.abs(sc - o64).max()
    assert d_hip <= max(4.0 * d_ref, 2e-3), (d_hip, d_ref)
    assert (sc.argmax(1) == o64.argmax(1)).mean() >= 0.96


def test_deferred_final_pass_is_bit_identical():
    """run_batch(defer_final=True): the final pass + GNN of batch i runs on a third stream while batch i+1 is ingested and
    adapted (double-buffered adapted weights / final-pass images).  Three consecutive batches must give exactly the
    synchronous results."""
    sd = synthetic.gnnnet_state_dict(seed=51)
    batches = [[synthetic.test_episode(1000 + 10 * b + i, 5, 5, 15, 84, gen_examples=1) for i in range(2)] for b in range(3)]
    rs = np.random.RandomState(19)
    perms = [[[rs.permutation(100)] for _ in range(2)] for _ in range(3)]
    e0 = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
    ref = [e0.run_batch(batches[b], perms=perms[b]).clone() for b in range(3)]
    e1 = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
    got = [e1.run_batch(batches[b], perms=perms[b], defer_final=True) for b in range(3)]
    torch.cuda.synchronize()
    for b in range(3):
        assert torch.equal(got[b], ref[b]), b


def test_prefetched_next_batch_is_bit_identical():
    """run_batch(prefetch=next): the next batch's ingest + stem cache run on a fourth stream beside the current inner loop
    (alternate support store / stem cache / final-pass store).  Four consecutive batches -- the third call is handed a list
    other than the one that was prefetched, which must be discarded -- give exactly the synchronous results."""
    sd = synthetic.gnnnet_state_dict(seed=53)
    batches = [[synthetic.test_episode(1100 + 10 * b + i, 5, 5, 15, 84, gen_examples=1) for i in range(2)] for b in range(4)]
    rs = np.random.RandomState(23)
    perms = [[[rs.permutation(100)] for _ in range(2)] for _ in range(4)]
    e0 = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
    ref = [e0.run_batch(batches[b], perms=perms[b]).clone() for b in range(4)]
    e1 = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=1, episodes_per_batch=2, device=DEV)
    nxt = [batches[1], batches[3], batches[3], None]          # call 1 prefetches batch 3 but batch 2 arrives: re-ingested
    got = [e1.run_batch(batches[b], perms=perms[b], defer_final=True, prefetch=nxt[b]) for b in range(4)]
    torch.cuda.synchronize()
    assert e1._pre_bufs is not None                           # the alternate buffers exist, i.e. the prefetch path ran
    for b in range(4):
        assert torch.equal(got[b], ref[b]), b


def test_finetune_linear_frozen_backbone_vs_reference_golden(golden_dir):
    """finetune_linear(freeze_backbone=True) (finetune.py:45-174, frozen branch): eval-mode features + 100 Adam steps on the
    Linear head in one launch (mft_linear_head_adam_run) against the scores the REFERENCE produced for the same episode,
    weights, classifier initialisation and numpy seed (G12); the numpy stream must end at the same position."""
    from meta_fine_tuning_amd import finetune as ft
    import argparse
    g = _g(golden_dir, "g12_finetune_linear_frozen.npz")
    sd = synthetic.gnnnet_state_dict_with_running_stats(seed=57)
    liz = synthetic.test_episode(97, 5, 5, 15, 84, gen_examples=1)
    ft.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=1)
    np.random.seed(10)
    sc = ft.finetune_linear([v.to(DEV) for v in liz], None, sd, None, linear=True, freeze_backbone=True,
                            classifier=(g["w0"], g["b0"])).cpu().numpy()
    assert sc.shape == (75, 5)
    assert np.abs(sc - g["scores"]).max() < 1e-3 and (sc.argmax(1) == g["scores"].argmax(1)).mean() >= 0.98, np.abs(sc - g["scores"]).max()
    assert np.array_equal(np.random.permutation(7), g["next_perm"])


def test_engine_full_config_vs_reference_golden(golden_dir):
    """BASELINE configs[1] exactly (fine_tune_epoch 5, 17 augmented views = 500 Adam steps) on one episode: engine scores
    against the scores the REFERENCE's finetune() produced (G5b) -- inside the Adam sign-flip envelope of SURVEY.md D7 -- and
    against the float64 oracle no further than 4x the reference's own fp32-vs-fp64 distance."""
    g = _g(golden_dir, "g5b_finetune_full.npz")
    sd = synthetic.gnnnet_state_dict(seed=13)
    liz = synthetic.test_episode(41 + 17, 5, 5, 15, 84, gen_examples=17)
    e = eng.FinetuneEngine(sd, n_views=19, fine_tune_epoch=5, episodes_per_batch=2, device=DEV)
    np.random.seed(10)
    st = np.random.get_state()
    sc = e.run_batch([liz])[0].cpu().numpy()
    ref = g["scores_E5_G17"]
    err = np.abs(sc - ref)
    assert err.max() < 5e-2 and (sc.argmax(1) == ref.argmax(1)).mean() >= 0.96, (err.max(), (sc.argmax(1) == ref.argmax(1)).mean())
    np.random.set_state(st)
    o64 = O.finetune_episode(sd, liz, 5, 5, total_epoch=5, dtype=torch.float64).numpy()
    assert np.abs(sc - o64).max() <= max(4.0 * np.abs(ref - o64).max(), 2e-3), (np.abs(sc - o64).max(), np.abs(ref - o64).max())
    e.close()


def test_inner_loop_500_steps_teacher_forced_trajectory(golden_dir):
    """The whole inner loop of the README setting (500 Adam steps over 19 views) on the index order of golden G4b: last-block
    weight norms after 105 and 500 steps against the reference's fp32 AND fp64 runs (the two differ by the Adam sign-flip
    envelope; the engine must sit in the same envelope), probe features against fp64 within 4x the reference's own distance."""
    g = _g(golden_dir, "g4b_inner_loop_long.npz")
    gs = _g(golden_dir, "g4c_inner_loop_spread.npz")
    size = 84
    sd = synthetic.resnet10_state_dict(seed=9)
    views = synthetic.test_episode(31, 5, 5, 15, size, gen_examples=17)
    full = {"feature." + k: v for k, v in sd.items()}
    full.update(synthetic.gnn_head_state_dict(seed=1))
    order = g["order"]
    for n_epochs, tag in ((None, 105), (5, 500)):
        # epochs of 100 steps each: 105 steps = one epoch + 5 steps -> run 2 epochs' permutations but stop the tables at 105
        e = eng.FinetuneEngine(full, n_views=19, fine_tune_epoch=5, episodes_per_batch=1, device=DEV)
        e._ingest([views], False)
        e.adapt.reset(e.W)
        e.prepare_batch()
        perms = [[order[ep * 500:(ep + 1) * 500] for ep in range(5)]]
        tables = e.step_tables(perms, 1)[:tag]
        e.inner_loop(tables)
        torch.cuda.synchronize()
        w = e.adapt.w.export(0)
        for key, gk in (("trunk.7.C1.weight", "wn_c1"), ("trunk.7.C2.weight", "wn_c2"), ("trunk.7.shortcut.weight", "wn_sc")):
            n_hip = float(w[key].norm())
            n32, n64 = float(g["%s_s%d_f32" % (gk, tag)]), float(g["%s_s%d_f64" % (gk, tag)])
            # the envelope comes from the REFERENCE's own fp32 variants (golden G4c: the same 500 steps re-run by the reference at
            # 1 / 2 / 4 / 8 ATen threads and with oneDNN off -- other summation orders of the same arithmetic): after 500 steps they
            # land 88.74 .. 89.17 on trunk.7.C1 against 89.37 in fp64.  105 steps: 4x the reference's fp32-vs-fp64 distance (the
            # standing gate, SURVEY.md D7 ii); 500 steps: 2x the reference's WORST fp32 variant's distance to fp64.
            spread = max(abs(float(gs["%s_s%d_%s" % (gk, tag, v)]) - n64) for v in gs["variants"])
            tol = max(4.0 * abs(n32 - n64), 0.02) if tag == 105 else max(2.0 * spread, 0.02)
            assert abs(n_hip - n64) <= tol, (key, tag, n_hip, n32, n64)
        # probe: the first five support images through the adapted network (train-mode BatchNorm, one group of 5)
        xa = torch.cat([v[:, :5].contiguous().view(25, 3, size, size) for v in [views[0]] + views], 0)
        feat = Fn.resnet10_forward(e.W, ops.nchw_to_nhwc(xa[:5].to(DEV)), Fn.Arena(DEV), ipg=5, slab=e.adapt.w).cpu().numpy()
        p32, p64 = g["probe_s%d_f32" % tag], g["probe_s%d_f64" % tag]
        d_ref = np.abs(p32 - p64)
        d_hip = np.abs(feat - p64)
        assert np.percentile(d_hip, 99) <= max(4.0 * np.percentile(d_ref, 99), 5e-3), (tag, np.percentile(d_hip, 99), np.percentile(d_ref, 99))
        e.close()


@pytest.mark.parametrize("ns,fname,ep_seed,marks", [(20, "g4d_inner_loop_2000.npz", 331, (500, 2000)),
                                                    (50, "g4e_inner_loop_5000.npz", 332, (2000, 5000))])
def test_inner_loop_full_length_teacher_forced_trajectory(golden_dir, ns, fname, ep_seed, marks):
    """The inner loop at the README length of the 20-shot (BASELINE configs[2]: 5 epochs x 400 mini-batches = 2000 Adam steps) and
    50-shot (configs[4], finetune_50.py:264-299: 5 x 1000 = 5000 steps) runs, on the index order of goldens G4d / G4e: last-block
    weight norms at two marks against the reference's fp64 run, inside the envelope the reference's OWN fp32 variants span
    around it (default, 1 ATen thread, oneDNN off -- other summation orders of the same arithmetic), probe features within 4x
    the reference's own fp32-vs-fp64 distance."""
    import os
    if not os.path.exists(os.path.join(golden_dir, fname)):
        pytest.skip("golden %s not generated" % fname)
    g = _g(golden_dir, fname)
    size = 84
    sd = synthetic.resnet10_state_dict(seed=9)
    views = synthetic.test_episode(ep_seed, 5, ns, 15, size, gen_examples=17)
    full = {"feature." + k: v for k, v in sd.items()}
    full.update(synthetic.gnn_head_state_dict(seed=1))
    order = g["order"]
    fp32_variants = [v for v in g["variants"] if v != "f64"]
    nt = 100 * ns                                            # 5 * ns supports x (19 + 1) views
    for tag in marks:
        e = eng.FinetuneEngine(full, n_support=ns, n_views=19, fine_tune_epoch=5, episodes_per_batch=1, device=DEV)
        e._ingest([views], False)
        e.adapt.reset(e.W)
        e.prepare_batch()
        perms = [[order[ep * nt:(ep + 1) * nt] for ep in range(5)]]
        e.inner_loop(e.step_tables(perms, 1)[:tag])
        torch.cuda.synchronize()
        w = e.adapt.w.export(0)
        for key, gk in (("trunk.7.C1.weight", "wn_c1"), ("trunk.7.C2.weight", "wn_c2"), ("trunk.7.shortcut.weight", "wn_sc")):
            n_hip, n64 = float(w[key].norm()), float(g["%s_s%d_f64" % (gk, tag)])
            spread = max(abs(float(g["%s_s%d_%s" % (gk, tag, v)]) - n64) for v in fp32_variants)
            assert abs(n_hip - n64) <= max(2.0 * spread, 0.02), (key, tag, n_hip, n64, spread)
        xa = views[0][:, :ns].contiguous().view(5 * ns, 3, size, size)           # (the probe = the first five support images)
        feat = Fn.resnet10_forward(e.W, ops.nchw_to_nhwc(xa[:5].to(DEV)), Fn.Arena(DEV), ipg=5, slab=e.adapt.w).cpu().numpy()
        p64 = g["probe_s%d_f64" % tag]
        d_ref = max(np.percentile(np.abs(g["probe_s%d_%s" % (tag, v)] - p64), 99) for v in fp32_variants)
        d_hip = np.percentile(np.abs(feat - p64), 99)
        assert d_hip <= max(4.0 * d_ref, 5e-3), (tag, d_hip, d_ref)
        e.close()


def test_adam_slab_placement_changes_no_result(monkeypatch):
    """engine.AdaptState._place picks the w / m / v (and second w) buffers among separately allocated candidates by the measured rate
    of the Adam-shaped stream over them (where the slabs live decides how fast they stream; it cannot change a result): the scores
    of a batch are bit-identical with the selection on and off, the report names four distinct candidates and its chosen triple is
    not slower than the first three allocations."""
    sd = synthetic.gnnnet_state_dict(seed=3)
    eps = [synthetic.test_episode(70 + i, 5, 5, 15, 84, gen_examples=1) for i in range(6)]
    perms = [[np.random.RandomState(40 + i).permutation(100)] for i in range(6)]
    outs = {}
    for k in ("12", "0"):
        monkeypatch.setenv("MFT_SLAB_CANDIDATES", k)
        e = eng.FinetuneEngine(sd, n_views=3, fine_tune_epoch=1, episodes_per_batch=6, device=DEV)
        outs[k] = e.run_batch(eps, perms=perms).clone()
        rep = e.adapt.placement
        if k == "0":
            assert rep is None
        else:
            assert rep["candidates"] >= 5 and len(set(rep["chosen"])) == 4
            # (a triple confirmed from an earlier process's hint is re-measured: its fresh rate may sit a little above the recorded best)
            assert rep["chosen_gbs"] >= 0.95 * rep["first_three_allocations_gbs"]      # (rates measured minutes apart: 2-3 % of noise)
            assert 0.97 * rep["worst_gbs"] <= rep["chosen_gbs"] <= 1.03 * rep["best_gbs"], rep
        e.close()
    assert torch.equal(outs["12"], outs["0"])
