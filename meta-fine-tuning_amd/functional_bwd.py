"""Taped forward + hand-written backward of the meta-training path, as sequences of HIP launches.

Covers what ``loss.backward()`` differentiates in MetaTemplate.train_loop2 / train_loop_finetune
(meta_template.py:76-109): the whole ResNet10 (one BatchNorm mini-batch per call, backbone.py:251-261,
401-439) and the GnnNet head (fc + BatchNorm1d + GNN_nl, gnnnet.py:30,76-87,210-217; gnn.py:16-166).
One episode per call as the reference's loops run it -- or k episodes in lockstep (``groups`` / ``episodes``: every BatchNorm keeps
per-episode statistics, the parameter gradients are the sums over the k episodes); gradients come back in the reference's layouts.
"""
import torch

from . import functional as Fn
from . import ops
from . import settings

L = ops._lib
RELU, LRELU, NONE = ops.ACT_RELU, ops.ACT_LRELU, ops.ACT_NONE
PAIR_RK_ROWS = settings.current().pair_rk_rows    # pair rows up to which Wcompute layers take the register-K form (0: never)
GEMM_RK_ROWS = settings.current().gemm_rk_rows    # linear layers of at most this many rows take the skinny register-K GEMM (0: never)
WGRAD_BATCH = settings.current().wgrad_batch      # deferred multi-problem weight gradients (ops.WgradBatch)


def _zeros(shape, dev):
    return torch.zeros(shape, device=dev, dtype=torch.float32)


def _empty(shape, dev):
    return torch.empty(shape, device=dev, dtype=torch.float32)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class ZeroLease:
    """Scratch buffers with ZERO padding that nobody re-fills: ``take(shape, dev)`` hands out a buffer that was zeroed when it was
    first allocated and whose users only ever write its valid columns (the padding columns feed data-gradient / weight-gradient
    reductions against zero weight rows, so they must be 0, not merely ignored).  A lease belongs to one tape (one forward + its
    backward); when the tape dies the buffers go back to the module-level free list and the next step takes them again -- the
    13 torch fill launches per meta-training step of round 5 are gone.  Buffers taken while a hipGraph is being recorded stay
    with that graph for good (its replays keep writing them; an eager step must never be handed the same memory)."""

    _free = {}

    def __init__(self):
        self.held = []
        self.pinned = _capturing()

    def take(self, shape, dev):
        key = (tuple(shape), dev.index if isinstance(dev, torch.device) else dev)
        lst = ZeroLease._free.get(key)
        t = lst.pop() if lst else torch.zeros(shape, device=dev, dtype=torch.float32)
        if _capturing():
            self.pinned = True
        self.held.append((key, t))
        return t

    def __del__(self):
        try:
            if not self.pinned:
                for key, t in self.held:
                    ZeroLease._free.setdefault(key, []).append(t)
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass


def bn_stats(x2d, C, rows, running=None, groups=1):
    """``rows`` = all rows of x2d; ``groups`` BatchNorm mini-batches of rows / groups rows each (k episodes in lockstep)."""
    rm, rv, nbt = running if running is not None else (None, None, None)
    return ops.bn_stats(x2d, C, rows // groups, groups, rm, rv, num_batches_tracked=nbt)


def bn_bwd(x2d, dy2d, C, rows, mean, rstd, gamma, y_act=None, act=NONE, need_dx=True, lease=None, groups=1, dbias_zero=None):
    """-> dx (or None), dgamma [C], dbeta [C] (summed over the ``groups`` mini-batches, which share the affine parameters).
    ``dbias_zero`` [C]: receives the (identically zero) gradient of a bias added in front of this BatchNorm."""
    dev = x2d.device
    dx = None
    if need_dx:          # padding columns (ld > C) must stay zero: they feed dgrad reductions against zero weight rows
        if x2d.shape[-1] == C:
            dx = _empty(x2d.shape, dev)
        else:
            dx = lease.take(x2d.shape, dev) if lease is not None else _zeros(x2d.shape, dev)
    dg, db = _empty((C,), dev), _empty((C,), dev)
    rpg = rows // groups
    ws = _empty((int(L.lib().mft_bn_backward_ws_floats(C, rpg, groups)),), dev)
    one = groups == 1                     # one group: its sums ARE the parameter gradients; several: summed by the finalize launch
    rc = L.lib().mft_bn_backward_act(ops._p(x2d), x2d.shape[-1], ops._p(dy2d), dy2d.shape[-1], ops._p(y_act),
                                     0 if y_act is None else y_act.shape[-1], ops._p(dx), 0 if dx is None else dx.shape[-1],
                                     C, rpg, groups, ops._p(mean), ops._p(rstd), ops._p(gamma), 0, ops._p(dg) if one else None,
                                     ops._p(db) if one else None, act, ops.LRELU_SLOPE, ops._p(ws), None if one else ops._p(dg),
                                     None if one else ops._p(db), ops._p(dbias_zero), ops._stream())
    L.check(rc, "mft_bn_backward_act")
    return dx, dg, db


def act_backward(dy, y, dx, C, act, accumulate, dy_off=0, y_off=0, dx_off=0):
    """dx[:, dx_off:dx_off+C] (+)= dy[:, dy_off:..] * act'(y[:, y_off:..]) on 2-D tensors with arbitrary row strides."""
    rows = dy.shape[0]
    rc = L.lib().mft_act_backward(dy.data_ptr() + 4 * dy_off, dy.shape[1], y.data_ptr() + 4 * y_off, y.shape[1],
                                  dx.data_ptr() + 4 * dx_off, dx.shape[1], C, rows, act, ops.LRELU_SLOPE,
                                  1 if accumulate else 0, ops._stream())
    L.check(rc, "mft_act_backward")


def colsum(x2d, C):
    rows = x2d.shape[0]
    out = _empty((C,), x2d.device)
    ws = _empty((((rows + 255) // 256) * C,), x2d.device)
    L.check(L.lib().mft_colsum(ops._p(x2d), x2d.shape[1], C, rows, ops._p(out), ops._p(ws), ops._stream()), "mft_colsum")
    return out


# =========================================================================================== ResNet10

def resnet10_forward_taped(W, x, running=None, groups=1):
    """x [n,H,W,3] NHWC, ``groups`` BatchNorm mini-batches of n / groups consecutive images (1 = the reference's loop: one episode per
    call; k = k episodes in lockstep, each with its own statistics, as k ranks of an episode-parallel run would have).
    Returns (features [n,512], tape)."""
    n = x.shape[0]
    assert n % groups == 0
    dev = x.device
    t = {"x": x, "n": n, "groups": groups}

    def run(name):
        return None if running is None else running.get(name)

    def conv3x3(name, inp, cin, cout, stride, rows_out):
        w3 = W.train_planes(name, cin, cout, 3, stride, rows_out)
        # the data-gradient operand's planes are registered HERE, on the caller's thread: the backward runs on autograd's device
        # thread, where the host-to-device copy of a new job table aborts the process on this stack (measured, round 5)
        W.train_planes(name, cin, cout, 3, stride, rows_out, transposed=True)
        if w3 is not None:                       # bf16x3: fp32-accurate on the 16-bit matrix cores (planes refreshed by W.repack())
            return ops.conv2d_x3(inp, w3, cout, 3, 3, stride, 1)
        return ops.conv2d(inp, W.conv[name], cout, 3, 3, stride, 1)

    c0 = ops.conv2d(x, W.conv["trunk.0"], 64, 7, 7, 2, 3)
    H0 = c0.shape[1]
    m0, s0 = bn_stats(c0.view(-1, 64), 64, n * H0 * H0, run("trunk.1"), groups)
    g0, b0 = W.bn["trunk.1"]
    PH = (H0 + 2 - 3) // 2 + 1
    a0 = _empty((n, PH, PH, 64), dev)
    arg = torch.empty((n, PH, PH, 64), device=dev, dtype=torch.uint8)
    L.check(L.lib().mft_bn_relu_maxpool_arg(ops._p(c0), ops._p(a0), ops._p(arg), n, H0, H0, 64, n // groups, ops._p(m0), ops._p(s0),
                                            ops._p(g0), ops._p(b0), ops._stream()), "mft_bn_relu_maxpool_arg")
    t.update(c0=c0, m0=m0, s0=s0, a0=a0, arg=arg)
    a = a0
    blocks = []
    for idx in (4, 5, 6, 7):
        cin, cout, stride = Fn.STAGES[idx]
        p = "trunk.%d" % idx
        H = a.shape[1]
        OH = (H + 2 - 3) // stride + 1
        rows = n * OH * OH
        b = {"p": p, "x": a, "cin": cin, "cout": cout, "stride": stride, "rows": rows}
        c1 = conv3x3(p + ".C1", a, cin, cout, stride, rows)
        g1, be1 = W.bn[p + ".BN1"]
        small = ops.bn_forward_small_ok(cout, rows // groups)       # (<= 512 rows per group: not at 84 x 84; smaller inputs only)
        if small:
            r1, m1, s1 = ops.bn_forward_small(c1.view(-1, cout), cout, rows // groups, groups, g1, be1, act=RELU, running=run(p + ".BN1"))
            r1 = r1.view(n, OH, OH, cout)
        else:
            m1, s1 = bn_stats(c1.view(-1, cout), cout, rows, run(p + ".BN1"), groups)
            r1 = ops.bn_apply(c1.view(-1, cout), cout, rows // groups, groups, m1, s1, g1, be1, act=RELU).view(n, OH, OH, cout)
        c2 = conv3x3(p + ".C2", r1, cout, cout, 1, rows)
        g2, be2 = W.bn[p + ".BN2"]
        if cin != cout and small:
            # BN2 + BNshortcut + add + ReLU of a small block exit: ONE launch (both statistics taken in its first pass)
            sc = ops.conv2d(a, W.conv[p + ".shortcut"], cout, 1, 1, stride, 0)
            gs, bs = W.bn[p + ".BNshortcut"]
            out, m2, s2, ms, ss = ops.bn_forward_small(c2.view(-1, cout), cout, rows // groups, groups, g2, be2, act=RELU,
                                                       running=run(p + ".BN2"), res=sc.view(-1, cout),
                                                       res_bn=(gs, bs, run(p + ".BNshortcut")))
            out = out.view(n, OH, OH, cout)
            b.update(sc=sc, ms=ms, ss=ss)
            b.update(c1=c1, m1=m1, s1=s1, r1=r1, c2=c2, m2=m2, s2=s2, out=out)
            blocks.append(b)
            a = out
            continue
        if cin != cout and WGRAD_BATCH:
            # BN2 and BNshortcut normalise two tensors that exist together: ONE statistics launch pair for both (mft_bn_stats_multi)
            sc = ops.conv2d(a, W.conv[p + ".shortcut"], cout, 1, 1, stride, 0)
            (m2, s2), (ms, ss) = ops.bn_stats_multi([(c2.view(-1, cout), cout, rows // groups, groups) + (run(p + ".BN2") or (None, None, None)),
                                                     (sc.view(-1, cout), cout, rows // groups, groups) + (run(p + ".BNshortcut") or (None, None, None))])
            gs, bs = W.bn[p + ".BNshortcut"]
            out = ops.bn_apply(c2.view(-1, cout), cout, rows // groups, groups, m2, s2, g2, be2, act=RELU, res=sc.view(-1, cout),
                               res_bn=(ms, ss, gs, bs)).view(n, OH, OH, cout)
            b.update(sc=sc, ms=ms, ss=ss)
            b.update(c1=c1, m1=m1, s1=s1, r1=r1, c2=c2, m2=m2, s2=s2, out=out)
            blocks.append(b)
            a = out
            continue
        m2, s2 = bn_stats(c2.view(-1, cout), cout, rows, run(p + ".BN2"), groups)
        if cin != cout:
            sc = ops.conv2d(a, W.conv[p + ".shortcut"], cout, 1, 1, stride, 0)
            ms, ss = bn_stats(sc.view(-1, cout), cout, rows, run(p + ".BNshortcut"), groups)
            gs, bs = W.bn[p + ".BNshortcut"]
            out = ops.bn_apply(c2.view(-1, cout), cout, rows // groups, groups, m2, s2, g2, be2, act=RELU, res=sc.view(-1, cout),
                               res_bn=(ms, ss, gs, bs)).view(n, OH, OH, cout)
            b.update(sc=sc, ms=ms, ss=ss)
        else:
            out = ops.bn_apply(c2.view(-1, cout), cout, rows // groups, groups, m2, s2, g2, be2, act=RELU,
                               res=a.view(-1, cin)).view(n, OH, OH, cout)
        b.update(c1=c1, m1=m1, s1=s1, r1=r1, c2=c2, m2=m2, s2=s2, out=out)
        blocks.append(b)
        a = out
    t["blocks"] = blocks
    feat = ops.global_avgpool(a)
    return feat, t


def resnet10_backward(W, t, dfeat, need):
    """Gradients of every ResNet10 parameter named in ``need`` (set of 'trunk.*' keys) in reference layouts."""
    n, groups = t["n"], t.get("groups", 1)
    dev = dfeat.device
    grads = {}
    wb = ops.WgradBatch(WGRAD_BATCH)            # every layer's weight gradient is registered as the pass goes and run at its end

    def dgrad3x3(name, dy, cin, cout, stride, rows, H_in):
        # registered by the forward (resnet10_forward_taped.conv3x3), never created here; the same size rule as the forward
        wt3 = W.train3.get((name, True)) if (Fn.TRAIN_X3 and rows >= Fn.TRAIN_X3_MIN_ROWS) else None
        if wt3 is not None:                      # stride 1: dx = conv(dy, tap-flipped channel-swapped weights), same padding
            return ops.conv2d_x3(dy, wt3, cin, 3, 3, 1, 1)
        return ops.conv2d_dgrad(dy, W.conv[name], cin, 3, 3, 1, stride=stride, in_hw=(H_in, H_in))

    last = t["blocks"][-1]
    d_out = ops.avgpool_relu_backward(dfeat.contiguous(), last["out"])
    for b in reversed(t["blocks"]):
        p, cin, cout, stride, rows = b["p"], b["cin"], b["cout"], b["stride"], b["rows"]
        x_in, out = b["x"], b["out"]
        H_in = x_in.shape[1]
        o2 = out.view(-1, cout)
        d2 = d_out.view(-1, cout)
        g2 = W.bn[p + ".BN2"][0]
        dsc = None
        if cin != cout and WGRAD_BATCH:
            # the backward of BN2 and of BNshortcut read the same d2 / o2: ONE launch triple for both (mft_bn_backward_act_multi)
            gs = W.bn[p + ".BNshortcut"][0]
            (dc2, dg, db), (dsc, dgs, dbs) = ops.bn_backward_multi([
                (b["c2"].view(-1, cout), d2, o2, cout, rows, groups, b["m2"], b["s2"], g2, RELU),
                (b["sc"].view(-1, cout), d2, o2, cout, rows, groups, b["ms"], b["ss"], gs, RELU)])
            grads[p + ".BNshortcut.weight"], grads[p + ".BNshortcut.bias"] = dgs, dbs
        else:
            dc2, dg, db = bn_bwd(b["c2"].view(-1, cout), d2, cout, rows, b["m2"], b["s2"], g2, y_act=o2, act=RELU, groups=groups)
        grads[p + ".BN2.weight"], grads[p + ".BN2.bias"] = dg, db
        dc2 = dc2.view(out.shape)
        grads[p + ".C2.weight"] = wb.add(b["r1"], dc2, cout, 3, 3, 1, 1)
        dr1 = dgrad3x3(p + ".C2", dc2, cout, cout, 1, rows, out.shape[1])
        g1 = W.bn[p + ".BN1"][0]
        dc1, dg, db = bn_bwd(b["c1"].view(-1, cout), dr1.view(-1, cout), cout, rows, b["m1"], b["s1"], g1,
                             y_act=b["r1"].view(-1, cout), act=RELU, groups=groups)
        grads[p + ".BN1.weight"], grads[p + ".BN1.bias"] = dg, db
        dc1 = dc1.view(out.shape)
        grads[p + ".C1.weight"] = wb.add(x_in, dc1, cout, 3, 3, stride, 1)
        dx = dgrad3x3(p + ".C1", dc1, cin, cout, stride, rows, H_in)
        if cin != cout:
            if dsc is None:
                gs = W.bn[p + ".BNshortcut"][0]
                dsc, dg, db = bn_bwd(b["sc"].view(-1, cout), d2, cout, rows, b["ms"], b["ss"], gs, y_act=o2, act=RELU, groups=groups)
                grads[p + ".BNshortcut.weight"], grads[p + ".BNshortcut.bias"] = dg, db
            dsc = dsc.view(out.shape)
            grads[p + ".shortcut.weight"] = wb.add(x_in, dsc, cout, 1, 1, stride, 0)
            dxs = ops.conv2d_dgrad(dsc, W.conv[p + ".shortcut"], cin, 1, 1, 0, stride=stride, in_hw=(H_in, H_in))
            act_backward(dxs.view(-1, cin), dxs.view(-1, cin), dx.view(-1, cin), cin, NONE, True)
        else:
            act_backward(d2, o2, dx.view(-1, cin), cin, RELU, True)            # identity shortcut through relu2
        d_out = dx
    # stem: maxpool + relu backward, BatchNorm backward, 7x7 wgrad
    c0, a0 = t["c0"], t["a0"]
    H0 = c0.shape[1]
    d_bn0 = _empty(c0.shape, dev)
    L.check(L.lib().mft_maxpool_relu_backward(ops._p(d_out), ops._p(t["arg"]), ops._p(a0), ops._p(d_bn0), n, H0, H0, 64,
                                              ops._stream()), "mft_maxpool_relu_backward")
    g0 = W.bn["trunk.1"][0]
    dc0, dg, db = bn_bwd(c0.view(-1, 64), d_bn0.view(-1, 64), 64, n * H0 * H0, t["m0"], t["s0"], g0, groups=groups)
    grads["trunk.1.weight"], grads["trunk.1.bias"] = dg, db
    grads["trunk.0.weight"] = wb.add(t["x"], dc0.view(c0.shape), 64, 7, 7, 2, 3)
    wb.flush()
    return grads


# =========================================================================================== GNN head

def _pad_rows32(w_pk):
    """Weight pack with its row count padded to a multiple of 32 (zero rows): the dgrad reduction runs over Cout."""
    cout, k = w_pk.shape
    cp = ops.round_up(cout, 32)
    if cp == cout:
        return w_pk
    buf = getattr(w_pk, "rows32", None)          # GnnHeadWeights packs into a row-padded buffer: nothing to build per step
    if buf is not None and buf.shape == (cp, k) and buf.data_ptr() == w_pk.data_ptr():
        return buf
    out = torch.zeros((cp, k), device=w_pk.device, dtype=torch.float32)
    out[:cout] = w_pk
    return out


def _linear_fwd(h, K, w, b, cout, lease=None):
    """h [rows, >=K] -> raw output [rows, roundup(cout,32)] (extra columns stay zero: they are dgrad/wgrad padding)."""
    rows = h.shape[0]
    ld = ops.round_up(cout, 32)
    if ld == cout:
        o = _empty((rows, ld), h.device)
    else:
        o = lease.take((rows, ld), h.device) if lease is not None else _zeros((rows, ld), h.device)
    _gemm_fwd(h, K, w, cout, b, o)
    return o


def _gemm_fwd(h, K, w, cout, b, out=None):
    """Forward GEMM of a head linear layer: the skinny register-K kernel for the few hundred rows of a meta-training step (one or a
    few episodes), the tile kernel otherwise."""
    if 0 < h.shape[0] <= GEMM_RK_ROWS and w.dim() == 2 and K <= 512 and w.shape[1] == K:
        return ops.gemm_rk(h, K, w, cout, bias=b, out=out)
    return ops.gemm(h, K, w, cout, bias=b, out=out)


def _linear_bwd(h, K, w, d_o, cout, need_dx=True, k_valid=0, db=None, wb=None):
    """-> (dx [rows,K] or None, dW [cout, k_valid or K] in nn.Linear's layout, db [cout]).  d_o [rows, roundup(cout,32)] with zero
    padding columns; ``k_valid``: the input features that are real (the rest of K is zero padding of the operand); ``db``: the bias
    gradient when the caller already has it (a bias in front of a BatchNorm: exact zeros from the BatchNorm-backward launch)."""
    rows = h.shape[0]
    cp = d_o.shape[1]
    hin = h.view(rows, 1, 1, K) if h.shape[1] == K else _narrow(h, K)
    wg = ops.conv2d_wgrad_oihw if wb is None else wb.add        # (wb: deferred to the end of the pass, ops.WgradBatch)
    dW = wg(hin, d_o.view(rows, 1, 1, cp), cp, 1, 1, 1, 0, cin_valid=k_valid, cout_valid=cout).view(cout, k_valid or K)
    if db is None:
        db = colsum(d_o, cp)[:cout]
    dx = None
    if need_dx:
        dx = _linear_dgrad(d_o, w, K)
    return dx, dW, db


def _linear_dgrad(d_o, w, K):
    """dx [rows, K] = d_o [rows, cp] @ W.  With a transposed operand beside the forward pack (GnnHeadWeights: ``w.wT``) the FORWARD GEMM
    kernel computes it (conflict-free 16-byte fragment reads instead of the BT form's 4-byte ones: 10-27 % per launch, bit-identical);
    else the data-gradient form reading the forward pack."""
    rows, cp = d_o.shape
    wT = getattr(w, "wT", None)
    if wT is not None and WGRAD_BATCH and wT.shape == (K, cp):
        return ops.gemm(d_o, cp, wT, K)
    return ops.conv2d_dgrad(d_o.view(rows, 1, 1, cp), _pad_rows32(w), K, 1, 1, 0).view(rows, K)


def _narrow(h, K):
    """[rows, ld] with ld > K -> contiguous [rows,1,1,K] copy (wgrad reads rows of exactly Cin floats)."""
    return h[:, :K].contiguous().view(h.shape[0], 1, 1, K)


PAIR_CHUNK_ROWS = 16384         # pair rows whose |x_i - x_j| exist at one time in the backward of layer 1 (<= 16 MB at F = 229: one
                                # chunk for two 5-shot episodes in lockstep, two for four)


def wcompute_taped(G, name, x, F, n_graphs, N, groups=1):
    """gnn.Wcompute.forward (gnn.py:78-132) on the fused per-pair kernels (csrc/pair_mlp.hip), keeping what the backward needs:
    the RAW layer outputs on the N(N+1)/2 upper-triangle pair rows [rows, 192 | 192 | 96 | 96] and every BatchNorm's (scale, shift,
    mean, rstd) [groups, C] -- ``groups`` episodes of n_graphs / groups graphs each, every episode with its own statistics.
    The pair tensor |x_i - x_j| [B*N*N, F] is never formed, here or in the backward."""
    layers, (w5, b5) = G.wc[name]
    lib = L.lib()
    dev = x.device
    P = N * (N + 1) // 2
    rows = n_graphs * P
    gpg = n_graphs // groups
    ij = Fn.pair_index_table(N, dev)
    Kp = ops.round_up(F, 32)
    # small problems (one or a few 5-shot episodes): 32-row tiles with the whole K in registers (csrc/pair_mlp.hip, register-K
    # form) instead of 128-row tiles that leave three quarters of the chip idle; fp32 MFMA only
    rk = rows <= PAIR_RK_ROWS and Kp <= 256 and not Fn.PAIR_F16X2
    tiles_m = int(lib.mft_pair_mlp_tiles_m_rk(gpg, N) if rk else lib.mft_pair_mlp_tiles_m(gpg, N))
    ws_mean, ws_m2, ws_n = (_empty((groups * tiles_m * 192,), dev), _empty((groups * tiles_m * 192,), dev),
                            _empty((groups * tiles_m,), dev))
    t = {"name": name, "F": F, "Kp": Kp, "ij": ij, "z": [], "bn": [], "groups": groups}
    h_in, ld_in, K, Kpad = x, x.shape[1], F, Kp
    sc_prev = sh_prev = None
    for li, (w, b, gam, beta, cout) in enumerate(layers):
        z = _empty((rows, cout), dev)
        sc, sh, m, s = (_empty((groups, cout), dev) for _ in range(4))
        if rk:
            L.check(lib.mft_pair_mlp_layer_rk(ops._p(h_in), ld_in, 0 if li == 0 else 1, ops._p(ij), ops._p(sc_prev), ops._p(sh_prev), ops._p(w),
                                              K, Kpad, ops._p(b), ops._p(z), cout, groups, gpg, N, ops.LRELU_SLOPE, ops._p(ws_mean),
                                              ops._p(ws_m2), ops._p(ws_n), ops._stream()), "mft_pair_mlp_layer_rk")
        else:
            L.check(lib.mft_pair_mlp_layer(ops._p(h_in), ld_in, 0 if li == 0 else 1, ops._p(ij), ops._p(sc_prev), ops._p(sh_prev), ops._p(w),
                                           K, Kpad, ops._p(b), ops._p(z), cout, groups, gpg, N, ops.LRELU_SLOPE, ops._p(ws_mean),
                                           ops._p(ws_m2), ops._p(ws_n), 1 if Fn.PAIR_F16X2 else 0, ops._stream()), "mft_pair_mlp_layer")
        L.check((lib.mft_pair_mlp_stats_finalize_rk if rk else lib.mft_pair_mlp_stats_finalize)(
            ops._p(ws_mean), ops._p(ws_m2), ops._p(ws_n), groups, tiles_m, cout, ops._p(gam), ops._p(beta), ops.BN_EPS, ops._p(sc),
            ops._p(sh), ops._p(m), ops._p(s), ops._stream()), "mft_pair_mlp_stats_finalize")
        t["z"].append(z); t["bn"].append((sc, sh, m, s))
        h_in, ld_in, K, Kpad, sc_prev, sh_prev = z, cout, cout, cout, sc, sh
    s_ut = _empty((rows,), dev)
    L.check(lib.mft_pair_mlp_score(ops._p(h_in), layers[3][4], ops._p(sc_prev), ops._p(sh_prev), ops._p(w5), ops._p(b5), ops.LRELU_SLOPE,
                                   ops._p(s_ut), groups, gpg, N, ops._stream()), "mft_pair_mlp_score")
    A = _empty((n_graphs, N, N), dev)
    L.check(lib.mft_masked_softmax_ut(ops._p(s_ut), ops._p(A), n_graphs, N, ops._stream()), "mft_masked_softmax_ut")
    t["A"] = A
    return A, t


def pair_activations(G, tapes, rows):
    """h_l = leaky_relu(BatchNorm(z_l)) of EVERY pair-MLP layer of the given Wcompute tapes in one launch (the forward keeps only the
    raw z_l; the backward needs h_l as the weight-gradient operand of layer l + 1): twelve problems that depend on the tape only,
    so they run together in front of the head's backward chain instead of one launch each inside it.  Leaves ``t["act"]``."""
    jobs = []
    for t in tapes:
        layers = G.wc[t["name"]][0]
        groups = t.get("groups", 1)
        for li in range(4):
            _, _, gam, beta, cout = layers[li]
            _, _, m, s = t["bn"][li]
            z = t["z"][li]
            jobs.append((z, cout, rows // groups, groups, m, s, gam, beta, LRELU, _empty(z.shape, z.device)))
    outs = ops.bn_apply_multi(jobs)
    for i, t in enumerate(tapes):
        t["act"] = outs[4 * i:4 * i + 4]


def _pair_activation(t, li, layers, rows):
    """h_l = leaky_relu(BatchNorm(z_l)) of layer li as a [rows, C] operand of the generic GEMM launches (a transient)."""
    if "act" in t:
        return t["act"][li]
    _, _, gam, beta, cout = layers[li]
    _, _, m, s = t["bn"][li]
    z = t["z"][li]
    groups = t.get("groups", 1)
    return ops.bn_apply(z, cout, rows // groups, groups, m, s, gam, beta, act=LRELU, out=_empty(z.shape, z.device))


def wcompute_backward(G, t, dA, x, dX, n_graphs, N, grads, prefix, wb=None):
    """Accumulates d(x) into dX[:, :F]; writes parameter gradients into ``grads`` under ``prefix``.  Works on the forward's
    upper-triangle rows: a merged row carries the sum of the reference's (i, j) and (j, i) gradients (csrc/pair_mlp.hip, backward
    section); layer 1's input |x_i - x_j| is produced for PAIR_CHUNK_ROWS rows at a time."""
    layers, (w5, b5) = G.wc[t["name"]]
    lib = L.lib()
    dev = x.device
    P = N * (N + 1) // 2
    groups = t.get("groups", 1)
    rows, n_tot = n_graphs * P, (n_graphs // groups) * N * N            # n_tot: the positions ONE episode's BatchNorm averages over
    ij, F, Kp = t["ij"], t["F"], t["Kp"]
    ds = t["lease"].take((rows, 32), dev)              # column 0: gradient of the compact symmetric score (columns 1..31 stay 0)
    rd = _empty((n_graphs * N,), dev)
    db5 = _empty((1,), dev)                    # conv2d_last.bias shifts every logit of a softmax row alike: zero gradient, from this launch
    L.check(lib.mft_pair_softmax_ut_backward(ops._p(t["A"]), ops._p(dA), ops._p(ij), ops._p(rd), ops._p(ds), 32, n_graphs, N,
                                             ops._p(db5), ops._stream()), "mft_pair_softmax_ut_backward")
    h4 = _pair_activation(t, 3, layers, rows)
    dh, dW, _ = _linear_bwd(h4, 96, w5, ds, 1, db=db5, wb=wb)
    del h4
    grads[prefix + ".conv2d_last.weight"] = dW.view(1, 96, 1, 1)
    grads[prefix + ".conv2d_last.bias"] = db5
    for li in (3, 2, 1, 0):
        w, b, gam, beta, cout = layers[li]
        sc, sh, m, s = t["bn"][li]
        z = t["z"][li]
        sums, dpar, dbz = _empty((groups, 2 * cout), dev), _empty((2 * cout,), dev), _empty((cout,), dev)
        ws = _empty((groups * int(lib.mft_pair_bwd_stats_ws_floats(rows // groups, cout)),), dev)
        dz = _empty((rows, cout), dev)
        # (dbz: the 1x1 convolution's bias sits in front of the BatchNorm -- its gradient is identically zero and comes back as zeros
        # from this launch; round 5 spent two column-sum launches per layer on the rounding noise of sum(dz))
        L.check(lib.mft_pair_bn_act_backward(ops._p(dh), dh.shape[1], ops._p(z), cout, ops._p(sc), ops._p(sh), ops._p(m), ops._p(s),
                                             ops._p(gam), ops._p(ij), N, rows // groups, groups, n_tot, ops.LRELU_SLOPE, ops._p(ws),
                                             ops._p(sums), ops._p(dpar), ops._p(dbz), ops._p(dz), ops._stream()),
                "mft_pair_bn_act_backward")
        grads[prefix + ".bn_%d.weight" % (li + 1)], grads[prefix + ".bn_%d.bias" % (li + 1)] = dpar[cout:], dpar[:cout]      # (views of this layer's own buffer)
        grads[prefix + ".conv2d_%d.bias" % (li + 1)] = dbz
        if li > 0:
            K = layers[li - 1][4]
            hin = _pair_activation(t, li - 1, layers, rows)
            dh, dW, _ = _linear_bwd(hin, K, w, dz, cout, db=dbz, wb=wb)
            del hin
            grads[prefix + ".conv2d_%d.weight" % (li + 1)] = dW.view(cout, K, 1, 1)
        else:
            # layer 1: its input is |x_i - x_j|: generated chunk by chunk, multiplied, dropped
            dW = None
            wp = _pad_rows32(w)
            for r0 in range(0, rows, PAIR_CHUNK_ROWS):
                nr = min(PAIR_CHUNK_ROWS, rows - r0)
                d = _empty((nr, Kp), dev)
                L.check(lib.mft_pair_absdiff_ut(ops._p(x), x.shape[1], ops._p(ij), ops._p(d), Kp, F, N, r0, nr, ops._stream()),
                        "mft_pair_absdiff_ut")
                dzc = dz[r0:r0 + nr]
                wg = ops.conv2d_wgrad_oihw if wb is None else wb.add
                part = wg(d.view(nr, 1, 1, Kp), dzc.view(nr, 1, 1, cout), cout, 1, 1, 1, 0, cin_valid=F).view(cout, F)
                if dW is None:
                    dW = part
                elif wb is None:                         # (more than PAIR_CHUNK_ROWS pair rows) dW += part, on the device
                    act_backward(part, part, dW, F, NONE, True)
                else:                                    # ... once the deferred launch has produced both
                    wb.then(lambda part=part, dW=dW: act_backward(part, part, dW, F, NONE, True))
                dd = _linear_dgrad(dzc, w, Kp)
                L.check(lib.mft_pair_dx_gather(ops._p(x), x.shape[1], ops._p(dd), Kp, ops._p(dX), dX.shape[1], n_graphs, N, F, r0, nr,
                                               ops._stream()), "mft_pair_dx_gather")
            grads[prefix + ".conv2d_1.weight"] = dW.view(cout, F, 1, 1)


def gconv_taped(G, name, A, x, F, n_graphs, N, lease=None, groups=1, into_x=False):
    """gnn.Gconv.forward (gnn.py:134-166).  ``into_x``: the caller appends leaky_relu(output) to the node features (GNN_nl's loop,
    gnn.py:160-165): where the one-launch BatchNorm runs, it writes x[:, F : F + cout] itself and None is returned in place of the
    output (no copy launch)."""
    w, b, g, beta, cout = G.gc[name]
    rows = n_graphs * N
    ldy = ops.round_up(2 * F, 32)
    y = ops.graph_aggregate(A, x, F, ldy)
    o = _linear_fwd(y, ldy, w, b, cout, lease)
    t = {"name": name, "F": F, "y": y, "raw": o, "A": A, "lease": lease, "groups": groups}
    if g is not None:
        if ops.bn_forward_small_ok(cout, rows // groups):      # 480 node rows per episode: statistics + apply in one launch
            if into_x:
                _, m, s = ops.bn_forward_small(o, cout, rows // groups, groups, g, beta, act=LRELU, out=x, out_col=F)
                t["stats"] = (m, s)
                return None, t
            ob, m, s = ops.bn_forward_small(o, cout, rows // groups, groups, g, beta, act=NONE, out=_empty(o.shape, o.device))
        else:
            m, s = ops.bn_stats(o, cout, rows // groups, groups)
            ob = ops.bn_apply(o, cout, rows // groups, groups, m, s, g, beta, act=NONE, out=_empty(o.shape, o.device))
        t["stats"] = (m, s)
        return ob, t
    return o, t


def gconv_backward(G, t, d_o, x, dX, n_graphs, N, grads, prefix, first=False, wb=None, db=None):
    """d_o: gradient w.r.t. the Gconv output (after its BatchNorm when present), [rows, roundup(cout,32)].
    Accumulates into dX[:, :F] (``first``: overwrites them -- dX is not zero-filled); returns dA."""
    w, b, g, beta, cout = G.gc[t["name"]]
    rows = n_graphs * N
    F = t["F"]
    dbz = db                                      # (a caller that already holds the bias gradient: layer_last)
    if g is not None:
        m, s = t["stats"]
        dbz = _empty((cout,), x.device)          # fc.bias sits in front of the BatchNorm1d: zero gradient, written by its backward
        d_o, dg, dbt = bn_bwd(t["raw"], d_o, cout, rows, m, s, g, lease=t.get("lease"), groups=t.get("groups", 1), dbias_zero=dbz)
        grads[prefix + ".bn.weight"], grads[prefix + ".bn.bias"] = dg, dbt
    dy, dW, db = _linear_bwd(t["y"], t["y"].shape[1], w, d_o, cout, k_valid=2 * F, db=dbz, wb=wb)
    grads[prefix + ".fc.weight"] = dW
    grads[prefix + ".fc.bias"] = db
    dA = _empty((n_graphs, N, N), x.device)
    L.check(L.lib().mft_graph_aggregate_backward(ops._p(t["A"]), ops._p(x), x.shape[1], ops._p(dy), dy.shape[1],
                                                 ops._p(dX), dX.shape[1], ops._p(dA), n_graphs, N, F, 0 if first else 1, ops._stream()),
            "mft_graph_aggregate_backward")
    return dA


def head_forward_taped(G, feats, n_way, n_support, n_query, fold=False, episodes=1):
    """GnnNet.fc + graph assembly + GNN_nl + score gather, keeping what backward needs.  ``episodes`` = k episodes in lockstep
    (feats = their feature rows one episode after the other): every BatchNorm of the head (fc's BatchNorm1d, the Wcomputes'
    BatchNorm2d, the Gconvs' BatchNorm1d) takes its statistics per episode, exactly as k separate calls would."""
    rows = feats.shape[0]
    dev = feats.device
    k = episodes
    assert rows % k == 0
    lease = ZeroLease()
    t = {"feats": feats, "n_way": n_way, "ns": n_support, "nq": n_query, "fold": fold, "lease": lease, "episodes": k}
    z_raw = _gemm_fwd(feats, 512, G.fc_w, 128, G.fc_b)
    if ops.bn_forward_small_ok(128, rows // k):
        z, mz, sz = ops.bn_forward_small(z_raw, 128, rows // k, k, G.fc_g, G.fc_beta, act=NONE, out=_empty(z_raw.shape, dev))
    else:
        mz, sz = ops.bn_stats(z_raw, 128, rows // k, k)
        z = ops.bn_apply(z_raw, 128, rows // k, k, mz, sz, G.fc_g, G.fc_beta, act=NONE, out=_empty(z_raw.shape, dev))
    t.update(z_raw=z_raw, mz=mz, sz=sz)
    N = n_way * (n_support + 1)
    n_graphs = k * n_query
    x = ops.build_graph_nodes(z, k, n_way, n_support, n_query, ld=256, fold=fold)
    F = 128 + n_way
    t.update(x=x, N=N, n_graphs=n_graphs, wc=[], gc=[], Fs=[])
    for i in range(2):
        A, tw = wcompute_taped(G, "layer_w%d" % i, x, F, n_graphs, N, k)
        tw["lease"] = lease
        ob, tg = gconv_taped(G, "layer_l%d" % i, A, x, F, n_graphs, N, lease, k, into_x=True)
        if ob is not None:
            ops.copy_cols(ob, x, F, 48, act=LRELU)
        t["wc"].append(tw); t["gc"].append(tg); t["Fs"].append(F)
        F += 48
    A, tw = wcompute_taped(G, "w_comp_last", x, F, n_graphs, N, k)
    tw["lease"] = lease
    out, tg = gconv_taped(G, "layer_last", A, x, F, n_graphs, N, lease, k)
    t["wc"].append(tw); t["gc"].append(tg); t["Fs"].append(F)
    scores = ops.gather_query_scores(out, k, n_way, n_support, n_query)
    return scores, t


def head_backward(G, t, dscores):
    """-> (dfeats [rows,512], grads keyed 'fc.*' / 'gnn.*' in reference layouts)."""
    n_way, ns, nq, fold = t["n_way"], t["ns"], t["nq"], t["fold"]
    x, N, n_graphs, k = t["x"], t["N"], t["n_graphs"], t.get("episodes", 1)
    dev = x.device
    grads = {}
    rows = n_graphs * N
    lease = t["lease"]
    dX = _empty((rows, 256), dev)              # first written (all 229 feature columns) by layer_last's aggregate backward: no zero fill
    d_out = _empty((rows, 32), dev)
    db_last = _empty((n_way,), dev)            # layer_last.fc.bias: the column sums of dscores, from the same launch
    L.check(L.lib().mft_gather_query_scores_backward(ops._p(dscores.contiguous()), ops._p(d_out), 32, k, n_way, ns, nq,
                                                     ops._p(db_last), ops._stream()), "mft_gather_query_scores_backward")
    wb = ops.WgradBatch(WGRAD_BATCH)            # the head's 16 weight gradients: registered as the pass goes, run together at its end
    if WGRAD_BATCH:
        pair_activations(G, t["wc"], n_graphs * (N * (N + 1) // 2))
    dA = gconv_backward(G, t["gc"][2], d_out, x, dX, n_graphs, N, grads, "gnn.layer_last", first=True, wb=wb, db=db_last)
    wcompute_backward(G, t["wc"][2], dA, x, dX, n_graphs, N, grads, "gnn.w_comp_last", wb=wb)
    for i in (1, 0):
        F = t["Fs"][i]
        d_ob = lease.take((rows, 64), dev)                   # 48 outputs padded to 64 (dgrad reduction width; the padding stays 0)
        act_backward(dX, x, d_ob, 48, LRELU, False, dy_off=F, y_off=F)
        dA = gconv_backward(G, t["gc"][i], d_ob, x, dX, n_graphs, N, grads, "gnn.layer_l%d" % i, wb=wb)
        wcompute_backward(G, t["wc"][i], dA, x, dX, n_graphs, N, grads, "gnn.layer_w%d" % i, wb=wb)
    per = n_way * ((2 * ns if fold else ns) + nq)
    dz = _empty((k * per, 128), dev)
    L.check(L.lib().mft_build_graph_nodes_backward(ops._p(dX), 256, ops._p(dz), 128, k, n_way, ns, nq, 1 if fold else 0,
                                                   ops._stream()), "mft_build_graph_nodes_backward")
    dbz = _empty((128,), dev)                  # fc.0.bias sits in front of fc.1 (BatchNorm1d): zero gradient, from its backward launch
    dzr, dg, db = bn_bwd(t["z_raw"], dz, 128, k * per, t["mz"], t["sz"], G.fc_g, groups=k, dbias_zero=dbz)
    grads["fc.1.weight"], grads["fc.1.bias"] = dg, db
    dfeats, dW, dbias = _linear_bwd(t["feats"], 512, G.fc_w, dzr, 128, db=dbz, wb=wb)
    grads["fc.0.weight"], grads["fc.0.bias"] = dW, dbias
    wb.flush()
    return dfeats, grads
