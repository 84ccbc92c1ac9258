"""Device-side functional layer: ResNet10 forward, last-block backward and the GNN head expressed
as sequences of launches into libmft_hip.so (via ``ops``).  Everything is *grouped*: a call
processes ``n_groups`` independent BatchNorm mini-batches / episodes at once, optionally with
per-group last-block weights (the episode-batched inner loop).

Reference call sites restated: backbone.ResNet.forward / SimpleBlock.forward
(backbone.py:251-261,401-439), the autograd of the inner-loop loss (finetune.py:286-299;
gnnnet.py:168-177), gnn.GNN_nl / Wcompute / Gconv (gnn.py:16-166), GnnNet.fc and forward_gnn
(gnnnet.py:30,82-87,210-217).
"""
import os

import numpy as np

import torch

from . import ops
from . import settings

# opt-in: data gradient of trunk.7.C2 from inside its weight-gradient + Adam launch (csrc/wgrad_dgrad.hip: one pass over the
# weights instead of two).  Standalone 1.60-1.63 ms + 31 us (col2im + BN1 backward) vs 1.71-1.72 ms for the two passes, but with
# 3 workgroups x 48 KB of w/m/v in flight per CU (8x fewer, longer workgroups) it loses more beside the trunk stream than the
# 64x64-tile kernel does: 70.8 / 71.5 / 69.4 vs 71.0 / 71.4 / 71.1 episodes/s -- off by default.
FUSED_DGRAD = settings.current().fused_dgrad
FUSED_LAST_BLOCK = settings.current().fused_last_block   # conv + BatchNorm fusions of the adapted block (csrc/skinny.hip)
# opt-in: trunk activations travel pre-split into bf16x3 planes between the x3 convolutions.  Bit-identical, but measured
# SLOWER (66.5 vs 69.0 episodes/s): the convolutions are not bound by the in-loader split (161 vs 160 us standalone) while the
# producers write 1.5x the bytes -- kept for the record (tests/test_engine_gpu.py::test_presplit_activation_planes_are_bit_identical)
X3_PLANES = settings.current().x3_planes
X3_FUSED_STATS = settings.current().x3_fused_stats   # BatchNorm statistics from the bf16x3 convolution epilogue
# frozen trunk blocks without helper launches: statistics stay per-tile partials that each consumer merges itself, BN1 + ReLU is
# applied by C2's loader, the stem's batch statistics are combined inside the pooled gather (12-13 launches per lockstep step
# instead of 24; MFT_X3_FOLD_BN=0 restores the separate finalize / apply / combine launches)
X3_FOLD_BN = settings.current().x3_fold_bn
# frozen trunk.4-6 convolutions on two fp16 pieces per operand (three products, csrc/conv_x3.hip "f16x2") instead of three bf16
# pieces (six products): half the matrix work at an error below an fp32-accumulating GEMM's.  Taken only when the state dict
# passes ``f16x2_safe`` (every operand provably inside fp16's range); MFT_TRUNK_F16X2=0 keeps the bf16x3 kernels.
TRUNK_F16X2 = settings.current().trunk_f16x2
TRAIN_X3 = settings.current().train_x3            # meta-training: large 3x3 layers on the bf16x3 kernels (ResNet10Weights.train_planes)
TRAIN_X3_MIN_ROWS = settings.current().train_x3_min_rows
# fused next-step forward (engine.fuse_next), opt-in variant: only trunk.7.C2's launch is the fused walking kernel, C1 / shortcut keep
# the plain gradient + Adam launches + one entry launch (last_block_backward).  Alone that chain is 100 us shorter, in situ it is
# slower: 87.2 vs 88.8-89.0 episodes/s (profiles/r04_d_fuse_c2_only_ab.txt) -- default 0 = all three layers fused
FUSE_NEXT_C2_ONLY = settings.current().fuse_next_c2_only
F16X2_BOUND = 3.0e4              # < 65504 with a factor 2 in hand
F16X2_FLOOR = 2.0 ** -10          # typical operand magnitude that keeps the pieces normal fp16 numbers
F16X2_MAX_ROWS = 1 << 20         # largest BatchNorm group (images x pixels) the bound is proven for (engine: 100 x 21 x 21 = 44,100)


def stem_rows_bound(ipg, image_size):
    """Upper bound of the largest BatchNorm group (the stem's: images x stem-output pixels) of a forward over ``ipg`` images per
    group, in the form ResNet10Weights.planes() re-derives it from a block's input side (odd sides round up: 84 -> 42 -> 21 ->
    11 gives 4 * 11 = 44 >= 42)."""
    oh = (image_size + 6 - 7) // 2 + 1
    ph = (oh + 2 - 3) // 2 + 1
    h6 = (ph + 2 - 3) // 2 + 1
    side = max(oh, 2 * ph, 4 * h6)
    return int(ipg) * side * side


def f16x2_safe(sd, prefix="", max_rows=None):
    """Can every operand of the frozen trunk.4-6 convolutions be split into fp16 pieces with full relative accuracy?
    Upper range -- weights: |w| < F16X2_BOUND; activations: each of these convolutions reads relu(BN(.)) or relu(BN(.) + BN(.)) /
    relu(BN(.) + x) (backbone.py:251-261), and a train-mode BatchNorm output over n rows is bounded by |gamma| sqrt(n - 1) + |beta|
    (a z-score cannot exceed sqrt(n - 1)); with n <= F16X2_MAX_ROWS that bound must stay below F16X2_BOUND for every input
    (reference initialisation, gamma = 1, beta = 0, backbone.py:11-16: 1024 / 2048).
    Lower range -- a piece below fp16's smallest normal number (6.1e-5) keeps an ABSOLUTE error of 2^-36 instead of a relative 2^-22:
    harmless beside O(1) operands, so the TYPICAL magnitude of every operand tensor must be well inside the normal range: rms weight
    and median |gamma| >= 2^-10 (He-initialised 3x3 weights: 0.03-0.06)."""
    import math
    max_rows = F16X2_MAX_ROWS if max_rows is None else int(max_rows)       # the largest BatchNorm group the caller will ever run

    def arr(key):              # numpy on purpose: the first torch CPU reduction of a process spins up the intra-op thread pool (1-2 s on a 256-core host)
        return np.abs(sd[prefix + key].detach().float().cpu().numpy())

    def g(name):
        w, b = arr(name + ".weight"), arr(name + ".bias")
        if not float(np.median(w)) >= F16X2_FLOOR:
            return float("inf")
        return float(w.max()) * math.sqrt(max_rows) + float(b.max())

    try:
        x_in = g("trunk.1")                                     # stem BatchNorm -> ReLU -> max pool -> trunk.4
        for idx, (cin, cout, _s) in STAGES.items():
            if idx == 7:
                break
            p = "trunk.%d" % idx
            for cname in (".C1", ".C2") + ((".shortcut",) if cin != cout else ()):
                w = arr(p + cname + ".weight")
                if not (float(w.max()) < F16X2_BOUND and float(np.sqrt(np.mean(np.square(w, dtype=np.float64)))) >= F16X2_FLOOR):
                    return False
            r1 = g(p + ".BN1")
            x_out = g(p + ".BN2") + (g(p + ".BNshortcut") if cin != cout else x_in)
            if not (x_in < F16X2_BOUND and r1 < F16X2_BOUND):  # inputs of C1 / shortcut, and of C2 (NaN compares False)
                return False
            x_in = x_out
        return True                                             # (trunk.6's output feeds the per-episode fp32 / bf16x3 kernels)
    except KeyError:
        return False

STAGES = {4: (64, 64, 1), 5: (64, 128, 2), 6: (128, 256, 2), 7: (256, 512, 2)}

# adaptable tensors of trunk.7 in reference named_parameters() order (finetune.py:236-252)
ADAPT_KEYS = [
    "trunk.7.C1.weight", "trunk.7.BN1.weight", "trunk.7.BN1.bias",
    "trunk.7.C2.weight", "trunk.7.BN2.weight", "trunk.7.BN2.bias",
    "trunk.7.shortcut.weight", "trunk.7.BNshortcut.weight", "trunk.7.BNshortcut.bias",
]
ADAPT_SHAPES = {
    "trunk.7.C1.weight": (512, 256, 3, 3), "trunk.7.C2.weight": (512, 512, 3, 3),
    "trunk.7.shortcut.weight": (512, 256, 1, 1),
    "trunk.7.BN1.weight": (512,), "trunk.7.BN1.bias": (512,), "trunk.7.BN2.weight": (512,),
    "trunk.7.BN2.bias": (512,), "trunk.7.BNshortcut.weight": (512,), "trunk.7.BNshortcut.bias": (512,),
}
ADAPT_NUMEL = sum(int(torch.Size(s).numel()) for s in ADAPT_SHAPES.values())      # 3,673,088


class Arena:
    """Named, shape-keyed persistent device buffers: static addresses (hipGraph-friendly), no per-step malloc."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = torch.empty(tuple(shape), device=self.device, dtype=dtype)
            self.bufs[key] = t
        return t

    def existing(self, name, shape, dtype=torch.float32):
        """A buffer some earlier launch WROTE under this name and shape: a miss is a KeyError, never a fresh (uninitialised)
        allocation -- for readers that reconstruct a producer's key (engine._trunk_running_ema)."""
        key = (name, tuple(shape), dtype)
        if key not in self.bufs:
            near = sorted(str(k) for k in self.bufs if k[0] == name)
            raise KeyError("arena has no buffer %r with shape %s%s" % (name, tuple(shape), "; same name, other shapes: " + ", ".join(near) if near else ""))
        return self.bufs[key]

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class ResNet10Weights:
    """Packed device copy of a backbone.ResNet10 state dict (keys 'trunk.*' under ``prefix``)."""

    def __init__(self, sd, device, prefix="", x3=False, f16x2=None, max_rows=None):
        """``x3``: also keep split planes of the frozen trunk.4-6 weights (csrc/conv_x3.hip: fp32-accurate convolution on the
        bf16 / fp16 matrix cores); used wherever these layers run with shared weights.  ``f16x2``: two fp16 planes (three
        products) instead of three bf16 planes (six); None = TRUNK_F16X2 and f16x2_safe(sd, max_rows).
        ``max_rows``: the largest train-mode BatchNorm group (images x stem-output pixels) the caller will run through these
        weights -- the fp16 range proof of f16x2_safe is made for THAT size (default F16X2_MAX_ROWS), and ``planes()`` hands the
        fp16 planes out only to calls inside it."""
        self.device = device
        self.conv = {}
        self.conv3 = {}
        self.f16x2_rows = F16X2_MAX_ROWS if max_rows is None else max(int(max_rows), 1)
        if f16x2 is None:
            f16x2 = bool(x3) and TRUNK_F16X2 and not X3_PLANES and f16x2_safe(sd, prefix, self.f16x2_rows)
        self.f16x2 = bool(f16x2)
        split = ops.split_weight_h2 if self.f16x2 else ops.split_weight_x3
        self.bn = {}
        self.plan = ops.PackPlan()          # sources that are live device tensors: repack() refreshes every packed copy in one launch
        self.train3 = {}                    # meta-training: (layer, transposed) -> bf16x3 planes, registered on first use (train_planes)
        self.tplan = ops.SplitPlan()

        def dev(t):
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        def conv(name):
            src = sd[prefix + name + ".weight"]
            w = dev(src)
            self.conv[name] = ops.pack_conv_weight(w)
            if w.data_ptr() == src.data_ptr():
                self.plan.add(w, self.conv[name])
            if x3 and name.startswith(("trunk.4", "trunk.5", "trunk.6")):
                self.conv3[name] = split(self.conv[name])

        def bn(name):
            self.bn[name] = (dev(sd[prefix + name + ".weight"]), dev(sd[prefix + name + ".bias"]))

        conv("trunk.0")
        bn("trunk.1")
        for idx, (cin, cout, s) in STAGES.items():
            p = "trunk.%d" % idx
            conv(p + ".C1"); bn(p + ".BN1"); conv(p + ".C2"); bn(p + ".BN2")
            if cin != cout:
                conv(p + ".shortcut"); bn(p + ".BNshortcut")

    _STEM_SCALE = {"trunk.4": 2, "trunk.5": 2, "trunk.6": 4}      # block input side -> (at most) stem-output side

    def planes(self, p, ipg, H, fixed=None, running=None):
        """The split weight planes block ``p`` may use for a call with ``ipg`` images per BatchNorm group on H x H inputs: the
        bf16x3 planes always (no range condition); the fp16 planes only where f16x2_safe's proof holds -- TRAIN-mode BatchNorm
        in front of every convolution (a z-score over n rows is bounded by sqrt(n - 1); eval-mode ``fixed`` / ``running``
        statistics bound nothing) and no BatchNorm group larger than the ``max_rows`` the proof was made for (the stem's group,
        ipg x stem-output pixels, is the largest).  Otherwise {}: the caller's fp32-MFMA kernels."""
        if not self.conv3 or not self.f16x2:
            return self.conv3
        if fixed is not None or running is not None:
            return {}
        side = self._STEM_SCALE.get(p, 8) * H
        return self.conv3 if ipg * side * side <= self.f16x2_rows else {}

    def repack(self):
        """The source parameters changed in place (optimizer.step): refresh every packed convolution weight with ONE launch (and
        the split planes of the layers the meta-training step runs on the split-precision kernels with a second one).
        BatchNorm weights / biases alias the parameters and need nothing.  Only valid when every convolution's source is a
        live device tensor (``can_repack``)."""
        assert self.can_repack()
        self.plan.run()
        self.tplan.run()

    def train_planes(self, name, cin, cout, k, stride, rows_out, transposed=False):
        """Meta-training (weights change every step, one 105-image BatchNorm group, round 5): the bf16x3 planes of 3x3 layer
        ``name`` -- of its tap-flipped / channel-swapped data-gradient operand with ``transposed`` -- if the layer is worth running
        on the split-precision kernels, else None.  Worth it = enough output rows to fill the machine with 128-row tiles
        (measured on 105 images of 84 x 84, profiles/r05_m_metatrain_x3.txt: 21 x 21 and 11 x 11 maps 1.4-1.7x the K-sliced fp32-MFMA
        launch, 6 x 6 and 3 x 3 maps slower); bf16x3 has no range condition, so no proof about the changing weights is needed.
        Planes are created at the first request and refreshed by ``repack()`` from then on."""
        if not TRAIN_X3 or k != 3 or rows_out < TRAIN_X3_MIN_ROWS:
            return None
        ci, co = (cout, cin) if transposed else (cin, cout)          # the kernel's view: ci input channels -> co output channels
        if ci % 32 != 0 or co % 64 != 0 or (transposed and stride != 1):
            return None
        key = (name, transposed)
        pl = self.train3.get(key)
        if pl is None:
            if not self.can_repack():
                return None
            pl = self.train3[key] = self.tplan.add(self.conv[name], cout, cin, k * k, transposed)
        return pl

    def can_repack(self):
        return len(self.plan.jobs) == len(self.conv) and not self.conv3


class LastBlockSlab:
    """``E`` copies of trunk.7's nine adaptable tensors, tensor-major inside ONE flat fp32 buffer
    (so Adam over all episodes is a single launch and every kernel sees a constant group stride).
    Conv weights are kept in the packed [Cout][KH][KW][Cin] layout."""

    ORDER = [("c1w", 512 * 2304), ("c2w", 512 * 4608), ("scw", 512 * 256),
             ("bn1g", 512), ("bn1b", 512), ("bn2g", 512), ("bn2b", 512), ("bnsg", 512), ("bnsb", 512)]
    KEY = {"c1w": "trunk.7.C1.weight", "c2w": "trunk.7.C2.weight", "scw": "trunk.7.shortcut.weight",
           "bn1g": "trunk.7.BN1.weight", "bn1b": "trunk.7.BN1.bias", "bn2g": "trunk.7.BN2.weight",
           "bn2b": "trunk.7.BN2.bias", "bnsg": "trunk.7.BNshortcut.weight", "bnsb": "trunk.7.BNshortcut.bias"}

    def __init__(self, E, device, zero=True, flat=None):
        self.E = E
        total = E * ADAPT_NUMEL
        if flat is not None:                      # a buffer chosen by the caller (engine.AdaptState: placement by measured stream rate)
            assert flat.numel() == total and flat.dtype == torch.float32 and flat.is_contiguous()
            self.flat = flat.zero_() if zero else flat
        else:
            self.flat = torch.zeros(total, device=device) if zero else torch.empty(total, device=device)
        off = 0
        for name, n in self.ORDER:
            v = self.flat[off:off + E * n]
            if name.endswith("w"):
                cout = 512
                v = v.view(E, cout, n // cout)
            else:
                v = v.view(E, n)
            setattr(self, name, v)
            off += E * n
        assert off == total

    def load_shared(self, W):
        """Every episode starts from the same checkpoint weights (finetune.py:185-198)."""
        self.c1w.copy_(W.conv["trunk.7.C1"].unsqueeze(0).expand_as(self.c1w))
        self.c2w.copy_(W.conv["trunk.7.C2"].unsqueeze(0).expand_as(self.c2w))
        self.scw.copy_(W.conv["trunk.7.shortcut"].unsqueeze(0).expand_as(self.scw))
        for nm, bnname in (("bn1", "trunk.7.BN1"), ("bn2", "trunk.7.BN2"), ("bns", "trunk.7.BNshortcut")):
            g, b = W.bn[bnname]
            getattr(self, nm + "g").copy_(g.unsqueeze(0).expand(self.E, -1))
            getattr(self, nm + "b").copy_(b.unsqueeze(0).expand(self.E, -1))

    def export(self, e):
        """Episode ``e``'s tensors as a reference-keyed dict (OIHW conv weights)."""
        out = {}
        for name, _ in self.ORDER:
            key = self.KEY[name]
            t = getattr(self, name)[e]
            if name.endswith("w"):
                out[key] = ops.unpack_conv_weight(t.contiguous(), ADAPT_SHAPES[key])
            else:
                out[key] = t.clone()
        return out


def _bn_stats4(arena, tag, x, ipg, groups, running=None, fixed=None):
    """Mini-batch statistics of x per group; ``fixed`` = (mean [1,C], rstd [1,C]) short-circuits them (eval-mode BatchNorm:
    running statistics, one group)."""
    if fixed is not None:
        return fixed
    n, H, W, C = x.shape
    mean = arena.get(tag + ".mean", (groups, C))
    rstd = arena.get(tag + ".rstd", (groups, C))
    rows = ipg * H * W
    nws = max(int(ops._lib.lib().mft_bn_stats_ws_floats(C, rows, groups)), 1)
    ws = arena.get("bn.ws", (max(nws, 1 << 16),)) if nws <= (1 << 16) else arena.get(tag + ".ws", (nws,))
    rm = rv = nbt = None
    if running is not None:
        rm, rv, nbt = running if len(running) == 3 else (running + (None,))
    rc = ops._lib.lib().mft_bn_stats(ops._p(x), C, C, rows, groups, ops.BN_EPS, ops._p(mean), ops._p(rstd), ops._p(ws),
                                     ops._p(rm), ops._p(rv), 0.1, ops._p(nbt), ops._stream())
    ops._lib.check(rc, "mft_bn_stats")
    return mean, rstd


class StemCache:
    """trunk.0 outputs of a set of resident images plus their per-image BatchNorm moments.

    finetune.py:263-291 re-runs the whole backbone on every mini-batch, so each support image passes the (frozen,
    mini-batch independent) stem convolution once per epoch.  The cache computes it once per image; trunk.1's
    mini-batch statistics are recombined exactly from per-image moments (csrc/bn.hip)."""

    def __init__(self, W, n_slots, H, device, chunk=8192, pooled=None):
        """``pooled`` (default: env MFT_STEM_POOLED, on): keep only the per-window maxima / minima of the raw stem output
        (csrc/bn.hip, exact: the BatchNorm affine is monotone per channel) -- half the memory of the full-resolution cache and a
        quarter to a half of the bytes per gather; the full-resolution form (``c0``) exists for the planes / test paths."""
        self.W = W
        self.n_slots = n_slots
        self.OH = (H + 6 - 7) // 2 + 1
        self.PH = (self.OH + 2 - 3) // 2 + 1
        cfg = settings.current()
        self.pooled = cfg.stem_pooled if pooled is None else bool(pooled)
        chunk = chunk if cfg.stem_chunk is None else cfg.stem_chunk
        self.chunk = min(chunk, n_slots)
        self.mean = torch.empty((n_slots, 64), device=device)
        self.m2 = torch.empty((n_slots, 64), device=device)
        self.fused_fill = bool(self.pooled and cfg.stem_fused_fill and H == 84)      # (csrc/stem.hip: stem_cache_kernel<84>)
        if self.pooled:
            self.c0 = None
            # one chunk of raw stem output: only the three-launch fill needs it
            self._buf = None if self.fused_fill else torch.empty((self.chunk, self.OH, self.OH, 64), device=device)
            self.pmax = torch.empty((n_slots, self.PH, self.PH, 64), device=device)
            self.pmin = torch.empty((n_slots, self.PH, self.PH, 64), device=device)
        else:
            self.c0 = torch.empty((n_slots, self.OH, self.OH, 64), device=device)

    def nbytes(self):
        ts = [self.mean, self.m2] + ([self._buf, self.pmax, self.pmin] if self.pooled else [self.c0])
        return sum(t.numel() * 4 for t in ts if t is not None)

    @staticmethod
    def bytes_needed(n_slots, H, pooled=None, chunk=8192):
        """What a StemCache over ``n_slots`` images of side H will allocate, without allocating it (the engine's admission test)."""
        cfg = settings.current()
        pooled = cfg.stem_pooled if pooled is None else bool(pooled)
        chunk = min(chunk if cfg.stem_chunk is None else cfg.stem_chunk, n_slots)
        oh = (H + 6 - 7) // 2 + 1
        ph = (oh + 2 - 3) // 2 + 1
        per = 2 * 64 + (2 * ph * ph * 64 if pooled else oh * oh * 64)
        fused = bool(pooled and cfg.stem_fused_fill and H == 84)
        return 4 * (n_slots * per + (chunk * oh * oh * 64 if (pooled and not fused) else 0))

    def fill(self, x_nhwc):
        """x_nhwc [n_slots,H,H,3] -> conv outputs + moments (chunked launches; M = chunk*OH*OW rows each)."""
        lib = ops._lib.lib()
        n = x_nhwc.shape[0]
        assert n == self.n_slots
        if self.fused_fill:
            # ONE launch: convolution + per-image moments + per-window (max, min); the full-resolution output never reaches HBM
            wpk = self.W.conv["trunk.0"]
            rc = lib.mft_stem_cache_fill(ops._p(x_nhwc), ops._p(wpk), wpk.shape[-1], n, x_nhwc.shape[1], x_nhwc.shape[2],
                                         ops._p(self.pmax), ops._p(self.pmin), ops._p(self.mean), ops._p(self.m2), ops._stream())
            if rc != ops._lib.MFT_EINVAL:
                ops._lib.check(rc, "mft_stem_cache_fill")
                return
            self.fused_fill = False                                   # outside the kernel's domain: the three launches from now on
            self._buf = torch.empty((self.chunk, self.OH, self.OH, 64), device=x_nhwc.device)
        for i in range(0, n, self.chunk):
            j = min(i + self.chunk, n)
            c0 = self._buf[:j - i] if self.pooled else self.c0[i:j]
            ops.conv2d(x_nhwc[i:j], self.W.conv["trunk.0"], 64, 7, 7, 2, 3, out=c0)
            ops._lib.check(lib.mft_bn_image_moments(ops._p(c0), 64, 64, self.OH * self.OH, j - i,
                                                    ops._p(self.mean[i:j]), ops._p(self.m2[i:j]), ops._stream()),
                           "mft_bn_image_moments")
            if self.pooled:
                ops._lib.check(lib.mft_pool_window_minmax(ops._p(c0), ops._p(self.pmax[i:j]), ops._p(self.pmin[i:j]), j - i,
                                                          self.OH, self.OH, 64, ops._stream()), "mft_pool_window_minmax")

    def gather(self, idx, n, m, s, g, b, ipg, out, planes=None):
        """BN + ReLU + MaxPool of images idx with the mini-batch statistics (m, s) -> out [n, PH, PH, 64]."""
        if self.pooled and planes is None:
            ops._lib.check(ops._lib.lib().mft_bn_relu_pooled_gather(ops._p(self.pmax), ops._p(self.pmin), ops._p(idx), ops._p(out), n,
                                                                    self.PH, self.PH, 64, ipg, ops._p(m), ops._p(s), ops._p(g),
                                                                    ops._p(b), ops._stream()), "mft_bn_relu_pooled_gather")
            return out
        assert self.c0 is not None, "this path needs the full-resolution stem cache (StemCache(pooled=False))"
        return ops.bn_relu_maxpool_gather(self.c0, idx, n, m, s, g, b, ipg, out=out, planes=planes)

    def gather_moments(self, idx, n, g, b, ipg, out, stats=None):
        """gather() with the mini-batch statistics combined from the cached per-image moments in the same launch."""
        assert self.pooled
        m, s = stats if stats is not None else (None, None)
        ops._lib.check(ops._lib.lib().mft_bn_relu_pooled_gather_moments(
            ops._p(self.pmax), ops._p(self.pmin), ops._p(idx), ops._p(out), n, self.PH, self.PH, 64, ipg, ops._p(self.mean),
            ops._p(self.m2), self.OH * self.OH, ops.BN_EPS, ops._p(g), ops._p(b), ops._p(m), ops._p(s), ops._stream()),
            "mft_bn_relu_pooled_gather_moments")
        return out


def resnet10_trunk(W, x, arena, ipg, upto=7, running=None, tag="t", stem=None, fixed=None):
    """trunk[0..upto-1] with shared (frozen) weights.  x [n,H,W,3] NHWC -> activation entering trunk[upto].
    ``running``: optional dict bn-name -> (running_mean, running_var) updated when a single group is run.
    ``stem`` = (StemCache, idx_i32): take trunk.0 outputs of images idx from the cache instead of x (x is ignored)."""

    def run(name):
        return None if running is None else running.get(name)

    def fix(name):
        return None if fixed is None else fixed[name]

    g, b = W.bn["trunk.1"]
    if stem is not None:
        cache, idx = stem
        n = idx.numel()
        groups = n // ipg
        PH = (cache.OH + 2 - 3) // 2 + 1
        fold = X3_FOLD_BN and cache.pooled and not X3_PLANES and running is None and fixed is None
        if not fold:
            m = arena.get(tag + ".bn0.mean", (groups, 64))
            s = arena.get(tag + ".bn0.rstd", (groups, 64))
            ops.bn_combine_moments(cache.mean, cache.m2, idx, cache.OH * cache.OH, ipg, groups, mean=m, rstd=s)
        if (X3_PLANES and cache.c0 is not None and X3_FUSED_STATS and upto == 7 and running is None and fixed is None and ipg * ((PH + 3) // 4) ** 2 >= 128
                and all(("trunk.%d.C1" % i) in W.conv3 for i in (4, 5, 6))):
            # frozen trunk on pre-split activations: each tensor between two bf16x3 convolutions is written once as three
            # bf16 planes by its producer (pool / BatchNorm-apply) instead of being re-split by every consumer tile
            ap = arena.get(tag + ".p0p", (3, n * PH * PH, 64), torch.int16)
            a = cache.gather(idx, n, m, s, g, b, ipg, arena.get(tag + ".p0", (n, PH, PH, 64)), planes=ap)
            shape = (n, PH, PH, 64)
            for bi in (4, 5, 6):
                cin, cout, stride = STAGES[bi]
                p = "trunk.%d" % bi
                mode = "f32" if bi == 6 else "planes"
                r = simple_block(W, p, a if bi == 4 else None, arena, ipg, cin, cout, stride, None, tag + "." + p, xp=ap,
                                 xshape=shape, out_mode=mode)
                if mode == "f32":
                    return r
                _, ap = r
                OHb = (shape[1] + 2 - 3) // stride + 1
                shape = (n, OHb, OHb, cout)
        if fold:
            a = cache.gather_moments(idx, n, g, b, ipg, arena.get(tag + ".p0", (n, PH, PH, 64)))
        else:
            a = cache.gather(idx, n, m, s, g, b, ipg, arena.get(tag + ".p0", (n, PH, PH, 64)))
    else:
        n = x.shape[0]
        groups = n // ipg
        c0 = ops.conv2d(x, W.conv["trunk.0"], 64, 7, 7, 2, 3, out=arena.get(tag + ".c0", (n, (x.shape[1] + 6 - 7) // 2 + 1, (x.shape[2] + 6 - 7) // 2 + 1, 64)))
        m, s = _bn_stats4(arena, tag + ".bn0", c0, ipg, groups, run("trunk.1"), fix("trunk.1"))
        a = ops.bn_relu_maxpool(c0, m, s, g, b, imgs_per_group=ipg)
    for idx in (4, 5, 6, 7):
        if idx >= upto:
            break
        cin, cout, stride = STAGES[idx]
        p = "trunk.%d" % idx
        a = simple_block(W, p, a, arena, ipg, cin, cout, stride, running, tag + "." + p, fixed=fixed)
    return a


def _bn_small(arena, tag, x1, g1, b1, rows, groups, C, gbs, out, x2=None, g2=None, b2=None, res=None, pooled=None, hw=0, stats=None,
              stats2=None):
    """stats + normalise (+ second normalised branch | + residual) + ReLU (+ global average pool) in one launch
    (groups of <= 64 rows, csrc/bn.hip).  Returns (mean1, rstd1, mean2, rstd2)."""
    m1, s1 = stats if stats is not None else (arena.get(tag + ".mean", (groups, C)), arena.get(tag + ".rstd", (groups, C)))
    m2 = s2 = None
    if x2 is not None:
        m2, s2 = stats2 if stats2 is not None else (arena.get(tag + ".mean2", (groups, C)), arena.get(tag + ".rstd2", (groups, C)))
    rc = ops._lib.lib().mft_bn_small_forward(ops._p(x1), C, ops._p(x2), C, ops._p(res), C, ops._p(out), C, C, rows, groups,
                                             ops._p(g1), ops._p(b1), ops._p(g2), ops._p(b2), gbs, ops._p(m1), ops._p(s1),
                                             ops._p(m2), ops._p(s2), ops.ACT_RELU, 0.0, ops.BN_EPS, ops._p(pooled), hw,
                                             ops._stream())
    ops._lib.check(rc, "mft_bn_small_forward")
    return m1, s1, m2, s2


def simple_block(W, p, x, arena, ipg, cin, cout, stride, running=None, tag="b", slab=None, tape=None, pooled=None, fixed=None,
                 xp=None, xshape=None, out_mode="f32", stats_out=None):
    """SimpleBlock.forward (backbone.py:251-261).  ``slab``: per-group parameters (LastBlockSlab) or None for W's.
    ``pooled``: optional [n, cout] buffer; filled with the global average pool of the block output when the fused
    small-group path applies (``tape['pooled']`` is then True and the caller skips its own pooling launch).
    ``xp``: the block input pre-split into bf16x3 planes (int16 [3, n*H*W, cin]); frozen x3 blocks then keep their internal
    activation in planes too, and ``out_mode`` ("f32" | "planes" | "both") selects the form(s) of the block output
    (returns ``out`` or ``(out, out_planes)``; ``x`` may be None when only its planes are needed, with ``xshape`` = its shape).
    ``stats_out`` (small-group slab path): {"bn1" | "bn2" | "bns": (mean, rstd) [groups, cout]} -- where the three BatchNorms'
    batch statistics are written instead of arena buffers (the caller advances the running statistics from them afterwards)."""
    n, H, Wd, _ = x.shape if x is not None else xshape
    groups = n // ipg
    OH = (H + 2 - 3) // stride + 1

    def run(name):
        return None if running is None else running.get(name)

    def fix(name):
        return None if fixed is None else fixed[name]

    if slab is None:
        c1w, c2w = W.conv[p + ".C1"], W.conv[p + ".C2"]
        (g1, b1), (g2, b2) = W.bn[p + ".BN1"], W.bn[p + ".BN2"]
        scw = W.conv.get(p + ".shortcut")
        gs, bs = W.bn.get(p + ".BNshortcut", (None, None))
        gbs = 0
    else:
        c1w, c2w, scw = slab.c1w, slab.c2w, slab.scw
        g1, b1, g2, b2, gs, bs = slab.bn1g, slab.bn1b, slab.bn2g, slab.bn2b, slab.bnsg, slab.bnsb
        gbs = cout
    wipg = ipg if slab is not None else 0
    w3 = W.planes(p, ipg, H, fixed, running) if slab is None else {}

    def conv(name, inp, wpk, k, s, pd, out):
        if (p + name) in w3:
            return ops.conv2d_x3(inp, w3[p + name], cout, k, k, s, pd, out=out)
        return ops.conv2d(inp, wpk, cout, k, k, s, pd, imgs_per_group=wipg, out=out)

    rows = ipg * OH * OH
    if slab is not None and rows <= 64 and cout % 64 == 0 and cin != cout and running is None:
        # adapted last block in the episode-batched loop: 3 fused launches instead of 10 around the three convolutions
        so = stats_out or {}
        c1 = arena.get(tag + ".c1", (n, OH, OH, cout))
        r1 = arena.get(tag + ".r1", (n * OH * OH, cout))
        sc = arena.get(tag + ".sc", (n, OH, OH, cout))
        rc = ops._lib.MFT_EINVAL
        # (one group = a single episode: the weight-streaming kernels would run on a handful of workgroups; the K-sliced
        #  implicit GEMM + one small BatchNorm launch is several times faster there)
        if FUSED_LAST_BLOCK and c1w.dim() == 3 and scw.dim() == 3 and groups > ops.SMALL_GROUPS and not so:
            # C1 + BatchNorm + ReLU and the shortcut convolution (which samples C1's centre tap) in one launch
            m1 = arena.get(tag + ".bn1.mean", (groups, cout))
            s1 = arena.get(tag + ".bn1.rstd", (groups, cout))
            rc = ops._lib.lib().mft_block_entry_small_forward(
                ops._p(x), x.shape[-1], ops._p(c1w), c1w.shape[1] * c1w.shape[2], ops._p(scw), scw.shape[1] * scw.shape[2],
                ops._p(c1), ops._p(r1), ops._p(sc), n, H, Wd, cin, cout, stride, ipg, ops._p(g1), ops._p(b1), gbs, ops._p(m1),
                ops._p(s1), ops.BN_EPS, ops._stream())
            if rc != ops._lib.MFT_EINVAL:
                ops._lib.check(rc, "mft_block_entry_small_forward")
        if rc == ops._lib.MFT_EINVAL:                      # outside the fused kernel's domain: three launches
            conv(".C1", x, c1w, 3, stride, 1, c1)
            m1, s1, _, _ = _bn_small(arena, tag + ".bn1", c1, g1, b1, rows, groups, cout, gbs, r1, stats=so.get("bn1"))
            conv(".shortcut", x, scw, 1, stride, 0, sc)
        r1 = r1.view(n, OH, OH, cout)
        c2 = arena.get(tag + ".c2", (n, OH, OH, cout))
        out = arena.get(tag + ".out", (n * OH * OH, cout))
        rc = ops._lib.MFT_EINVAL
        if FUSED_LAST_BLOCK and pooled is not None and c2w.dim() == 3 and groups > ops.SMALL_GROUPS and not so:
            # C2 + both BatchNorms + residual add + ReLU + global average pool in one launch
            m2, s2, ms, ss = (arena.get(tag + ".bn2." + k, (groups, cout)) for k in ("mean", "rstd", "mean2", "rstd2"))
            rc = ops._lib.lib().mft_block_exit_small_forward(
                ops._p(r1), ops._p(c2w), c2w.shape[1] * c2w.shape[2], ops._p(sc), ops._p(c2), ops._p(out), ops._p(pooled), n, OH,
                OH, cout, ipg, ops._p(g2), ops._p(b2), ops._p(gs), ops._p(bs), gbs, ops._p(m2), ops._p(s2), ops._p(ms),
                ops._p(ss), ops.BN_EPS, ops._stream())
            if rc != ops._lib.MFT_EINVAL:
                ops._lib.check(rc, "mft_block_exit_small_forward")
        if rc == ops._lib.MFT_EINVAL:
            conv(".C2", r1, c2w, 3, 1, 1, c2)
            m2, s2, ms, ss = _bn_small(arena, tag + ".bn2", c2, g2, b2, rows, groups, cout, gbs, out, x2=sc, g2=gs, b2=bs,
                                       pooled=pooled, hw=OH * OH, stats=so.get("bn2"), stats2=so.get("bns"))
        out = out.view(n, OH, OH, cout)
        if tape is not None:
            tape.update(x=x, c1=c1, m1=m1, s1=s1, r1=r1, c2=c2, m2=m2, s2=s2, sc=sc, ms=ms, ss=ss, out=out,
                        pooled=pooled is not None)
        return out
    # frozen shared-weight blocks on the bf16x3 kernels: the statistics of the BatchNorm behind each convolution come out
    # of the convolution's own epilogue (no separate pass over its output)
    fuse_stats = (slab is None and running is None and fixed is None and rows >= 128 and X3_FUSED_STATS)

    def conv_bn(name, bnname, inp, wpk, k, s, pd, obuf, stag):
        if fuse_stats and (p + name) in w3:
            H_in = inp.shape[1]
            nws = int(ops._lib.lib().mft_conv2d_x3_stats_ws_floats(n, H_in, H_in, cout, k, k, s, pd))
            ws = arena.get("x3.statws", (max(nws, 1 << 20),)) if nws <= (1 << 20) else arena.get(stag + ".statws", (nws,))
            mean = arena.get(stag + ".mean", (groups, cout))
            rstd = arena.get(stag + ".rstd", (groups, cout))
            r = ops.conv2d_x3_bnstats(inp, w3[p + name], cout, k, k, s, pd, ipg, obuf, ws, mean, rstd)
            if r is not None:
                return r
        o = conv(name, inp, wpk, k, s, pd, obuf)
        mm, ss_ = _bn_stats4(arena, stag, o, ipg, groups, run(p + bnname), fix(p + bnname))
        return o, mm, ss_

    if (X3_FOLD_BN and xp is None and fuse_stats and tape is None and out_mode == "f32"
            and all((p + nm) in w3 for nm in ((".C1", ".C2", ".shortcut") if cin != cout else (".C1", ".C2")))):
        r = _frozen_block_folded(W, p, x, arena, ipg, cin, cout, stride, tag)
        if r is not None:
            return r

    if xp is not None and fuse_stats and tape is None and all((p + nm) in w3 for nm in ((".C1", ".C2", ".shortcut") if cin != cout
                                                                                      else (".C1", ".C2"))):
        # pre-split activation chain: every x3 convolution reads bf16 planes written once by its producer
        def conv_bn_p(name, inp_p, h_in, k, s, pd, obuf, stag):
            nws = int(ops._lib.lib().mft_conv2d_x3_stats_ws_floats(n, h_in, h_in, cout, k, k, s, pd))
            ws = arena.get("x3.statws", (max(nws, 1 << 20),)) if nws <= (1 << 20) else arena.get(stag + ".statws", (nws,))
            mean = arena.get(stag + ".mean", (groups, cout))
            rstd = arena.get(stag + ".rstd", (groups, cout))
            return ops.conv2d_x3p_bnstats(inp_p, n, h_in, h_in, w3[p + name], cout, k, k, s, pd, ipg, obuf, ws, mean, rstd)

        c1, m1, s1 = conv_bn_p(".C1", xp, H, 3, stride, 1, arena.get(tag + ".c1", (n, OH, OH, cout)), tag + ".bn1")
        r1p = arena.get(tag + ".r1p", (3, n * OH * OH, cout), torch.int16)
        ops.bn_apply(c1.view(-1, cout), cout, rows, groups, m1, s1, g1, b1, act=ops.ACT_RELU, fma_affine=True, gb_group_stride=gbs, planes=r1p,
                     write_y=False)
        c2, m2, s2 = conv_bn_p(".C2", r1p, OH, 3, 1, 1, arena.get(tag + ".c2", (n, OH, OH, cout)), tag + ".bn2")
        want_f32 = out_mode in ("f32", "both")
        out = arena.get(tag + ".out", (n * OH * OH, cout)) if want_f32 else None
        outp = arena.get(tag + ".outp", (3, n * OH * OH, cout), torch.int16) if out_mode in ("planes", "both") else None
        if cin != cout:
            sc, ms, ss = conv_bn_p(".shortcut", xp, H, 1, stride, 0, arena.get(tag + ".sc", (n, OH, OH, cout)), tag + ".bns")
            res, res_bn = sc.view(-1, cout), (ms, ss, gs, bs)
        else:
            res, res_bn = x.view(-1, cin), None
        if outp is not None:
            ops.bn_apply(c2.view(-1, cout), cout, rows, groups, m2, s2, g2, b2, act=ops.ACT_RELU, fma_affine=True, res=res, res_bn=res_bn, out=out,
                         gb_group_stride=gbs, planes=outp, write_y=want_f32)
        else:
            ops.bn_apply(c2.view(-1, cout), cout, rows, groups, m2, s2, g2, b2, act=ops.ACT_RELU, fma_affine=True, res=res, res_bn=res_bn, out=out,
                         gb_group_stride=gbs)
        out = out.view(n, OH, OH, cout) if out is not None else None
        return out if out_mode == "f32" else (out, outp)
    assert out_mode == "f32" or xp is None or not fuse_stats, "planes output requested outside the pre-split chain"

    c1, m1, s1 = conv_bn(".C1", ".BN1", x, c1w, 3, stride, 1, arena.get(tag + ".c1", (n, OH, OH, cout)), tag + ".bn1")
    r1 = ops.bn_apply(c1.view(-1, cout), cout, rows, groups, m1, s1, g1, b1, act=ops.ACT_RELU, fma_affine=fuse_stats,
                      out=arena.get(tag + ".r1", (n * OH * OH, cout)), gb_group_stride=gbs).view(n, OH, OH, cout)
    c2, m2, s2 = conv_bn(".C2", ".BN2", r1, c2w, 3, 1, 1, arena.get(tag + ".c2", (n, OH, OH, cout)), tag + ".bn2")
    out = arena.get(tag + ".out", (n * OH * OH, cout))
    sc = ms = ss = None
    if cin != cout:
        sc, ms, ss = conv_bn(".shortcut", ".BNshortcut", x, scw, 1, stride, 0, arena.get(tag + ".sc", (n, OH, OH, cout)),
                             tag + ".bns")
        ops.bn_apply(c2.view(-1, cout), cout, rows, groups, m2, s2, g2, b2, act=ops.ACT_RELU, fma_affine=fuse_stats, res=sc.view(-1, cout),
                     res_bn=(ms, ss, gs, bs), out=out, gb_group_stride=gbs)
    else:
        ops.bn_apply(c2.view(-1, cout), cout, rows, groups, m2, s2, g2, b2, act=ops.ACT_RELU, fma_affine=fuse_stats, res=x.view(-1, cin),
                     out=out, gb_group_stride=gbs)
    out = out.view(n, OH, OH, cout)
    if tape is not None:
        tape.update(x=x, c1=c1, m1=m1, s1=s1, r1=r1, c2=c2, m2=m2, s2=s2, sc=sc, ms=ms, ss=ss, out=out)
    return out


_X3WS_FITS = {}


def _frozen_block_folded(W, p, x, arena, ipg, cin, cout, stride, tag):
    """SimpleBlock.forward (backbone.py:251-261) of a frozen shared-weight block in 3-5 launches: every convolution leaves its
    BatchNorm statistics as per-tile partials, C2's loader applies BN1 + ReLU, the exit launch merges BN2's (and the shortcut
    BatchNorm's) partials itself.  Returns None (nothing launched) when the first convolution is outside the bf16x3 domain."""
    n, H, _, _ = x.shape
    groups = n // ipg
    OH = (H + 2 - 3) // stride + 1
    rows = ipg * OH * OH
    lib = ops._lib.lib()
    fits = _X3WS_FITS.get((cout, rows))
    if fits is None:
        fits = _X3WS_FITS[(cout, rows)] = bool(lib.mft_bn_apply_x3ws_fits(cout, rows, 1))
    if not fits:
        return None                       # one BatchNorm batch of thousands of rows (the 100-image final pass): separate launches
    (g1, b1), (g2, b2) = W.bn[p + ".BN1"], W.bn[p + ".BN2"]
    w3 = W.conv3

    def partials(name, inp, k, s, pd, obuf, stag):
        ws = arena.get(stag + ".statws", (int(lib.mft_conv2d_x3_stats_ws_floats(n, inp.shape[1], inp.shape[1], cout, k, k, s, pd)),))
        r = ops.conv2d_x3_bnstats(inp, w3[p + name], cout, k, k, s, pd, ipg, obuf, ws, None, None)
        return None if r is None else ws

    c1 = arena.get(tag + ".c1", (n, OH, OH, cout))
    ws1 = partials(".C1", x, 3, stride, 1, c1, tag + ".bn1")
    if ws1 is None:
        return None
    c2 = arena.get(tag + ".c2", (n, OH, OH, cout))
    ws2 = arena.get(tag + ".bn2.statws", (int(lib.mft_conv2d_x3_stats_ws_floats(n, OH, OH, cout, 3, 3, 1, 1)),))
    if ops.conv2d_x3_bnin_bnstats(c1, ws1, g1, b1, w3[p + ".C2"], cout, ipg, c2, ws2) is None:
        # (scale, shift) table does not fit beside the tile (256 channels on 6x6 maps): BN1 as its own launch, statistics merged there
        r1 = ops.bn_apply_x3ws(c1.view(-1, cout), cout, rows, groups, ws1, g1, b1, arena.get(tag + ".r1", (n * OH * OH, cout)),
                               act=ops.ACT_RELU).view(n, OH, OH, cout)
        if ops.conv2d_x3_bnstats(r1, w3[p + ".C2"], cout, 3, 3, 1, 1, ipg, c2, ws2, None, None) is None:
            raise RuntimeError("frozen block %s: C2 outside the bf16x3 domain after C1 was inside it" % p)
    out = arena.get(tag + ".out", (n * OH * OH, cout))
    if cin != cout:
        gs, bs = W.bn[p + ".BNshortcut"]
        sc = arena.get(tag + ".sc", (n, OH, OH, cout))
        wss = partials(".shortcut", x, 1, stride, 0, sc, tag + ".bns")
        if wss is None:
            raise RuntimeError("frozen block %s: shortcut outside the bf16x3 domain after C1 was inside it" % p)
        ops.bn_apply_x3ws(c2.view(-1, cout), cout, rows, groups, ws2, g2, b2, out, act=ops.ACT_RELU, res=sc.view(-1, cout),
                          res_ws=wss, res_gamma=gs, res_beta=bs)
    else:
        ops.bn_apply_x3ws(c2.view(-1, cout), cout, rows, groups, ws2, g2, b2, out, act=ops.ACT_RELU, res=x.view(-1, cin))
    return out.view(n, OH, OH, cout)


def last_block_forward(W, a, arena, ipg, slab=None, tape=None, running=None, tag="f", fixed=None, stats_out=None):
    """trunk.7 + global average pool on the activation ``a`` [n,h,w,256] entering the last block -> features [n,512]."""
    n = a.shape[0]
    feat = arena.get(tag + ".feat", (n, 512))
    info = tape if tape is not None else {}
    out = simple_block(W, "trunk.7", a, arena, ipg, 256, 512, 2, running, tag + ".trunk.7", slab=slab, tape=info, pooled=feat,
                       fixed=fixed, stats_out=stats_out)
    if not info.get("pooled", False):
        ops._lib.check(ops._lib.lib().mft_global_avgpool(ops._p(out), ops._p(feat), n, out.shape[1] * out.shape[2], 512,
                                                         ops._stream()), "mft_global_avgpool")
    return feat


def resnet10_forward(W, x, arena, ipg=0, slab=None, tape=None, running=None, tag="f", fixed=None):
    """Full ResNet10(flatten=True) forward: x [n,H,W,3] NHWC -> features [n,512].  Train mode (mini-batch statistics per
    group of ``ipg`` images) unless ``fixed`` = {bn name: (mean [1,C], rstd [1,C])} supplies eval-mode statistics.
    ``slab`` selects per-group last-block parameters (episode-batched inner loop)."""
    n = x.shape[0]
    if ipg <= 0 or fixed is not None:
        ipg = n
    a = resnet10_trunk(W, x, arena, ipg, upto=7, running=running, tag=tag, fixed=fixed)
    return last_block_forward(W, a, arena, ipg, slab=slab, tape=tape, running=running, tag=tag, fixed=fixed)


def _dgrad_c2_bn1(lib, tape, params, grads, arena, tag, dc2, c1, r1, n, oh, ow, C, ipg, bn_bwd):
    """Data gradient of trunk.7.C2 followed by the BatchNorm1 + ReLU backward -> dc1 (separate pass over C2's weights)."""
    # dgrad of C2 must read the pre-update weights: it runs before the fused wgrad+Adam of C2
    rc = ops._lib.MFT_EINVAL
    if FUSED_LAST_BLOCK and n // ipg > ops.SMALL_GROUPS:
        # (one group -- a single episode, the meta-fine-tuning training loop: the weight-streaming kernel would run on 16 workgroups;
        #  the K-sliced implicit GEMM + the BatchNorm backward launches below take a sixth of its time)
        # data gradient of C2 with the BatchNorm1 + ReLU backward in its epilogue (dr1 is not materialised)
        dc1 = arena.get(tag + ".dc1", tuple(c1.shape))
        w2 = params.c2w
        rc = lib.mft_conv2d_dgrad_bn_backward_small(ops._p(dc2), C, ops._p(w2), ops._p(dc1), C, n, oh, ow, C, C, 3, 3, 1, ipg,
                                                    w2.shape[1] * w2.shape[2], ops._p(c1), ops._p(r1), ops._p(tape["m1"]),
                                                    ops._p(tape["s1"]), ops._p(params.bn1g), C, ops._p(grads.bn1g),
                                                    ops._p(grads.bn1b), ops._stream())
        if rc != ops._lib.MFT_EINVAL:
            ops._lib.check(rc, "mft_conv2d_dgrad_bn_backward_small")
    if rc == ops._lib.MFT_EINVAL:
        dr1 = ops.conv2d_dgrad(dc2, params.c2w, 512, 3, 3, 1, imgs_per_group=ipg, out=arena.get(tag + ".dr1", (n, oh, ow, C)))
        dc1 = bn_bwd(c1, dr1, tape["m1"], tape["s1"], params.bn1g, grads.bn1g, grads.bn1b, r1, "dc1")
    return dc1


def next_step_tape(arena, tag, n, oh, ow, C, groups):
    """Buffers the fused weight-gradient + Adam + next-forward launches (csrc/wgrad_fwd.hip) fill for inner step t+1: the same
    fields simple_block's tape carries for the adapted last block, plus the pooled feature."""
    rows = n * oh * ow
    t = {k: arena.get(tag + "." + k, (n, oh, ow, C)) for k in ("c1", "r1", "c2", "sc", "out")}
    for k in ("m1", "s1", "m2", "s2", "ms", "ss"):
        t[k] = arena.get(tag + "." + k, (groups, C))
    t["feat"] = arena.get(tag + ".feat", (n, C))
    t["pooled"] = True
    assert t["c1"].numel() == rows * C
    return t


def last_block_backward(tape, dfeat, params, grads, arena, ipg, tag="bw", adam=None, ce=None, nxt=None, before_next_read=None):
    """Backward of CE(feat) through avgpool + trunk.7 only (everything below is frozen: SURVEY §2.3 K13).
    ``params``/``grads``: LastBlockSlab (per-group).  Gradients are written in place into ``grads``.
    ``adam`` = (m_slab, v_slab, step, lr): fuse the Adam update of the three conv weights into the wgrad
    epilogues (their gradients are then not materialised) and update the BatchNorm affine tail separately.
    ``nxt`` = (x_next | None, tape_next | None) (needs ``adam``): the three weight-gradient + Adam launches also run the NEXT inner
    step's trunk.7 forward on ``x_next`` from the weight tiles they have just updated and fill ``tape_next`` (next_step_tape) --
    no forward launch reads the updated weights back (csrc/wgrad_fwd.hip).
    ``before_next_read``: a callable (the caller's stream wait for the producer of ``x_next``) that is run EXACTLY ONCE, right
    before the first launch that reads ``x_next``; the function asserts that it ran whenever ``x_next`` was read, and refuses the
    argument on paths that never read ``x_next`` (an unconsumed wait would be a silent race).  The BatchNorm-affine Adam tail then runs BEFORE them
    (their epilogues apply the updated gamma / beta), C1 and the shortcut before C2 (whose forward consumes r1 and sc of step
    t+1).  x_next None: the last inner step (update only)."""
    x, c1, r1, c2, sc, out = tape["x"], tape["c1"], tape["r1"], tape["c2"], tape["sc"], tape["out"]
    n, oh, ow, C = out.shape
    groups = n // ipg
    rows = ipg * oh * ow
    lib = ops._lib.lib()
    fused_ce = FUSED_LAST_BLOCK and ce is not None and C % 64 == 0 and ipg <= 16
    d_out = None
    if fused_ce:
        pass                                    # the loss gradient is formed inside the fused BatchNorm-backward launch below
    elif ce is not None:
        # ce = (feat, labels_i32, loss_out): cross entropy on the pooled feature and its gradient through AvgPool + ReLU in
        # one launch (dfeat is not materialised)
        d_out = arena.get(tag + ".dout", (n, oh, ow, C))
        feat, labels, loss = ce
        ops._lib.check(lib.mft_ce_pool_backward(ops._p(feat), ops._p(labels), ipg, groups, C, oh * ow, ops._p(out), ops._p(d_out),
                                                ops._p(loss), ops._stream()), "mft_ce_pool_backward")
    else:
        d_out = arena.get(tag + ".dout", (n, oh, ow, C))
        ops._lib.check(lib.mft_avgpool_relu_backward(ops._p(dfeat), ops._p(out), ops._p(d_out), n, oh * ow, C,
                                                     ops._stream()), "mft_avgpool_relu_backward")

    def bn_bwd(xraw, dy, mean, rstd, gamma, dgamma, dbeta, relu_out, name, need_dx=True):
        dx = arena.get(tag + "." + name, tuple(xraw.shape)) if need_dx else None
        rc = lib.mft_bn_backward(ops._p(xraw), C, ops._p(dy), C, ops._p(relu_out), C, ops._p(dx), C, C, rows, groups,
                                 ops._p(mean), ops._p(rstd), ops._p(gamma), C, ops._p(dgamma), ops._p(dbeta),
                                 ops._stream())
        ops._lib.check(rc, "mft_bn_backward")
        return dx

    def wgrad(xin, dy, name, k, stride, pad):
        if adam is None:
            ops.conv2d_wgrad(xin, dy, 512, k, k, stride, pad, imgs_per_group=ipg, out=getattr(grads, name))
        else:
            m, v, step, lr = adam
            hyper = step if torch.is_tensor(step) else None          # device-resident bias corrections (graph replay)
            ops.conv2d_wgrad_adam(xin, dy, getattr(params, name), getattr(m, name), getattr(v, name), 512, k, k,
                                  stride, pad, 1 if hyper is not None else step, imgs_per_group=ipg, lr=lr, hyper=hyper)

    if fused_ce:
        # CE + AvgPool/ReLU backward + both BatchNorm backwards of the residual join in one launch (d_out never reaches HBM)
        feat, labels, loss = ce
        dc2 = arena.get(tag + ".dc2", tuple(c2.shape))
        dsc = arena.get(tag + ".dsc", tuple(sc.shape))
        rc = lib.mft_ce_pool_bn_backward2(ops._p(feat), ops._p(labels), ipg, groups, C, oh * ow, ops._p(out), ops._p(c2), ops._p(sc),
                                          ops._p(dc2), ops._p(dsc), ops._p(tape["m2"]), ops._p(tape["s2"]), ops._p(params.bn2g),
                                          ops._p(tape["ms"]), ops._p(tape["ss"]), ops._p(params.bnsg), C, ops._p(grads.bn2g),
                                          ops._p(grads.bn2b), ops._p(grads.bnsg), ops._p(grads.bnsb), ops._p(loss), ops._stream())
        ops._lib.check(rc, "mft_ce_pool_bn_backward2")
    elif C % 64 == 0:
        dc2 = arena.get(tag + ".dc2", tuple(c2.shape))
        dsc = arena.get(tag + ".dsc", tuple(sc.shape))
        rc = lib.mft_bn_backward2(ops._p(c2), ops._p(sc), C, ops._p(d_out), C, ops._p(dc2), ops._p(dsc), C, C, rows, groups,
                                  ops._p(tape["m2"]), ops._p(tape["s2"]), ops._p(params.bn2g), ops._p(tape["ms"]),
                                  ops._p(tape["ss"]), ops._p(params.bnsg), C, ops._p(grads.bn2g), ops._p(grads.bn2b),
                                  ops._p(grads.bnsg), ops._p(grads.bnsb), ops._stream())
        ops._lib.check(rc, "mft_bn_backward2")
    else:
        dc2 = bn_bwd(c2, d_out, tape["m2"], tape["s2"], params.bn2g, grads.bn2g, grads.bn2b, None, "dc2")
        dsc = bn_bwd(sc, d_out, tape["ms"], tape["ss"], params.bnsg, grads.bnsg, grads.bnsb, None, "dsc")
    done_c2 = False
    if FUSED_DGRAD and FUSED_LAST_BLOCK and adam is not None and rows <= 64 and C % 64 == 0:
        # weight gradient + Adam of C2 that also forms the data gradient from the weight tiles it streams (one partial per tap),
        # then col2im + BatchNorm1/ReLU backward in a small launch: C2's weights are read once per backward instead of twice
        m_, v_, step, lr = adam
        hyper = step if torch.is_tensor(step) else None
        w2 = params.c2w
        dxp = arena.get(tag + ".dxp", (groups, 9, rows, C))
        dc1 = arena.get(tag + ".dc1", tuple(c1.shape))
        if ops.conv2d_wgrad_adam_dgrad(r1, dc2, w2, m_.c2w, v_.c2w, dxp, 1 if hyper is not None else step, ipg, lr=lr,
                                       hyper=hyper):
            ops._lib.check(lib.mft_col2im_bn_backward_small(ops._p(dxp), ops._p(c1), ops._p(r1), ops._p(dc1), n, oh, ow, C, ipg,
                                                            ops._p(tape["m1"]), ops._p(tape["s1"]), ops._p(params.bn1g), C,
                                                            ops._p(grads.bn1g), ops._p(grads.bn1b), ops._stream()),
                           "mft_col2im_bn_backward_small")
            done_c2 = True
    def adam_tail():
        m, v, step, lr = adam
        nb = params.E * 6 * 512                       # BatchNorm affine tail of the tensor-major slab
        hyper = step if torch.is_tensor(step) else None
        ops.adam_step(params.flat[-nb:], grads.flat[-nb:], m.flat[-nb:], v.flat[-nb:], 1 if hyper is not None else step,
                      lr=lr, hyper=hyper)

    if before_next_read is not None and (nxt is None or nxt[0] is None):
        raise RuntimeError("before_next_read given, but this call never reads x_next")
    if nxt is not None:
        assert adam is not None and not done_c2
        x_next, tn = nxt[0], nxt[1]
        waited = [before_next_read is None]

        def wait_next_once():
            if not waited[0]:
                before_next_read()
                waited[0] = True
        m_, v_, step, lr = adam
        hyper = step if torch.is_tensor(step) else None
        st = 1 if hyper is not None else step
        dc1 = _dgrad_c2_bn1(lib, tape, params, grads, arena, tag, dc2, c1, r1, n, oh, ow, C, ipg, bn_bwd)
        adam_tail()
        kw = dict(lr=lr, hyper=hyper)
        fw = x_next is not None
        ok = hybrid = False
        if FUSE_NEXT_C2_ONLY and FUSED_LAST_BLOCK and groups > ops.SMALL_GROUPS:
            # Only trunk.7.C2 takes the fused launch (opt-in).  Alone, C1's and the shortcut's fused launches are 154 + 50 us longer
            # than the plain gradient + Adam launches and save the 137 us entry launch; the entry forward of step t+1 then reads the
            # updated C1 / shortcut weights once more (1.3 of the 14.7 MB per episode).  In situ the trunk stream finishes no earlier
            # beside the plain launches than beside the walking kernel, and the step is 1.7 % slower.
            wgrad(x, dsc, "scw", 1, 2, 0)
            wgrad(x, dc1, "c1w", 3, 2, 1)
            hybrid = ok = True
            if fw:
                wait_next_once()
                rc = lib.mft_block_entry_small_forward(
                    ops._p(x_next), x_next.shape[-1], ops._p(params.c1w), params.c1w.shape[1] * params.c1w.shape[2], ops._p(params.scw),
                    params.scw.shape[1] * params.scw.shape[2], ops._p(tn["c1"]), ops._p(tn["r1"]), ops._p(tn["sc"]), n, x_next.shape[1],
                    x_next.shape[2], x_next.shape[-1], C, 2, ipg, ops._p(params.bn1g), ops._p(params.bn1b), C, ops._p(tn["m1"]),
                    ops._p(tn["s1"]), ops.BN_EPS, ops._stream())
                if rc == ops._lib.MFT_EINVAL:
                    raise RuntimeError("block entry launch outside its domain after next_forward_ok accepted the shape")
                ops._lib.check(rc, "mft_block_entry_small_forward")
        if not hybrid:
            if fw:
                wait_next_once()
            ok = ops.wgrad_adam_next_forward(x, dsc, params.scw, m_.scw, v_.scw, 1, 1, 2, 0, st, ipg, x_next=x_next, mode=ops.WF_RAW,
                                             raw=tn["sc"] if fw else None, **kw)
            ok = ok and ops.wgrad_adam_next_forward(x, dc1, params.c1w, m_.c1w, v_.c1w, 3, 3, 2, 1, st, ipg, x_next=x_next,
                                                    mode=ops.WF_ENTRY, raw=tn["c1"] if fw else None, act=tn["r1"] if fw else None,
                                                    gamma=params.bn1g, beta=params.bn1b, gbs=C, mean=tn["m1"] if fw else None,
                                                    rstd=tn["s1"] if fw else None, **kw)
        ok = ok and ops.wgrad_adam_next_forward(r1, dc2, params.c2w, m_.c2w, v_.c2w, 3, 3, 1, 1, st, ipg,
                                                x_next=tn["r1"] if fw else None, mode=ops.WF_EXIT, raw=tn["c2"] if fw else None,
                                                act=tn["out"] if fw else None, gamma=params.bn2g, beta=params.bn2b, gbs=C,
                                                mean=tn["m2"] if fw else None, rstd=tn["s2"] if fw else None,
                                                sc_raw=tn["sc"] if fw else None, gamma_s=params.bnsg, beta_s=params.bnsb,
                                                mean_s=tn["ms"] if fw else None, rstd_s=tn["ss"] if fw else None,
                                                pooled=tn["feat"] if fw else None, **kw)
        if not ok:
            raise RuntimeError("wgrad_adam_next_forward: shape outside the fused kernel's domain (the caller checks next_forward_ok)")
        assert waited[0] or not fw, "x_next was read without the caller's stream wait"
        if fw:
            tn["x"] = x_next
        return
    if not done_c2:
        dc1 = _dgrad_c2_bn1(lib, tape, params, grads, arena, tag, dc2, c1, r1, n, oh, ow, C, ipg, bn_bwd)
        wgrad(r1, dc2, "c2w", 3, 1, 1)
    wgrad(x, dc1, "c1w", 3, 2, 1)
    wgrad(x, dsc, "scw", 1, 2, 0)
    if adam is not None:
        adam_tail()


def next_forward_ok(ipg, H6):
    """Domain of the fused next-step forward for trunk.7 (3x3 / stride 2 on an H6 x H6 map): all of an episode's output pixels
    must fit the kernel's 48-pixel tile (84x84 inputs: 5 images x 3 x 3 = 45)."""
    oh = (H6 + 2 - 3) // 2 + 1
    return FUSED_LAST_BLOCK and not FUSED_DGRAD and 0 < ipg <= 8 and ipg * oh * oh <= 48


# ------------------------------------------------------------------------------------------ GNN head

class GnnHeadWeights:
    """Packed device copy of GnnNet.fc and GnnNet.gnn (state dict keys 'fc.*', 'gnn.*')."""

    def __init__(self, sd, device, n_way):
        self.n_way = n_way
        self.plan = ops.PackPlan()
        n_packed = [0]

        def dev(t):
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        def packed(src):
            w = dev(src)
            pk = ops.pack_conv_weight(w, rows32=True)       # (48-, 5- and 1-row matrices: zero rows up to 32 for the data gradient)
            n_packed[0] += 1
            if w.data_ptr() == src.data_ptr():
                self.plan.add(w, pk)
                # W^T (zero-padded to [Kpad, roundup(Cout, 32)]) beside it: the meta-training backward computes dx = dy @ W with the
                # forward GEMM kernel on this operand (round 6); refreshed with the forward packs by the one repack launch
                pk.wT = torch.zeros((pk.shape[1], ops.round_up(w.shape[0], 32)), device=device, dtype=torch.float32)
                self.plan.add_transposed(w, pk.wT)
                self.has_wT = True
            return pk

        self.fc_w = packed(sd["fc.0.weight"])
        self.fc_b = dev(sd["fc.0.bias"])
        self.fc_g, self.fc_beta = dev(sd["fc.1.weight"]), dev(sd["fc.1.bias"])
        self.wc = {}
        self.gc = {}
        for name in ("layer_w0", "layer_w1", "w_comp_last"):
            layers = []
            for li in range(1, 5):
                w = sd["gnn.%s.conv2d_%d.weight" % (name, li)]
                layers.append((packed(w), dev(sd["gnn.%s.conv2d_%d.bias" % (name, li)]),
                               dev(sd["gnn.%s.bn_%d.weight" % (name, li)]), dev(sd["gnn.%s.bn_%d.bias" % (name, li)]),
                               w.shape[0]))
            last = (packed(sd["gnn.%s.conv2d_last.weight" % name]),
                    dev(sd["gnn.%s.conv2d_last.bias" % name]))
            self.wc[name] = (layers, last)
        for name, bn in (("layer_l0", True), ("layer_l1", True), ("layer_last", False)):
            w = sd["gnn.%s.fc.weight" % name]
            g = b = None
            if bn:
                g, b = dev(sd["gnn.%s.bn.weight" % name]), dev(sd["gnn.%s.bn.bias" % name])
            self.gc[name] = (packed(w), dev(sd["gnn.%s.fc.bias" % name]), g, b, w.shape[0])
        self._n_packed = n_packed[0]
        if getattr(self, "has_wT", False):
            self.plan.run()                                  # (fills the transposed operands once; the forward packs are rewritten with the same values)

    def can_repack(self):
        return len(self.plan.jobs) == self._n_packed * (2 if getattr(self, "has_wT", False) else 1)

    def repack(self):
        """Refresh every packed weight from its (in-place updated) source parameter with one launch."""
        assert self.can_repack()
        self.plan.run()


_PAIR_IJ = {}
FUSED_PAIR_MLP = settings.current().fused_pair_mlp
PAIR_F16X2 = settings.current().pair_f16x2           # pair-MLP layers on the fp16 matrix cores as fp32-accurate two-piece products
PAIR_MLP_BYTES = int(settings.current().pair_mlp_gb * (1 << 30))     # raw-activation budget per chunk of episodes


def pair_index_table(N, device):
    """(i << 16) | j of the upper-triangle pair rows, i-major: p(i, j) = i*N - i(i-1)/2 + (j - i), i <= j."""
    key = (N, str(device))
    t = _PAIR_IJ.get(key)
    if t is None:
        import numpy as np
        i, j = np.triu_indices(N)
        t = torch.from_numpy(((i.astype(np.int64) << 16) | j).astype(np.int32)).to(device)
        _PAIR_IJ[key] = t
    return t


def wcompute(G, name, x, F, n_graphs, N, n_groups, arena, tag="wc"):
    """gnn.Wcompute.forward (gnn.py:78-132): x [n_graphs*N, ld] -> A [n_graphs, N, N].  BatchNorm statistics are
    per group of n_graphs/n_groups graphs (one episode), over all graphs*N*N pair positions.

    Fused form (csrc/pair_mlp.hip): five grid-wide phases (one per BatchNorm), each ONE launch over a chunk of episodes; the
    pair tensor |x_i - x_j| is generated in the first layer's loader, only the N(N+1)/2 pairs i <= j are computed (the score
    is symmetric), every layer writes its RAW output once and the next layer applies BatchNorm + leaky_relu while loading;
    statistics come out of the GEMM epilogues.  Episodes are processed in chunks so that the raw activations of a chunk stay
    within ``PAIR_MLP_BYTES`` whatever E is."""
    if not FUSED_PAIR_MLP:
        return wcompute_unfused(G, name, x, F, n_graphs, N, n_groups, arena, tag)
    layers, (w5, b5) = G.wc[name]
    lib = ops._lib.lib()
    gpg = n_graphs // n_groups
    P = N * (N + 1) // 2
    ij = pair_index_table(N, x.device)
    A = arena.get(tag + ".A", (n_graphs, N, N))
    widths = [l[4] for l in layers]                                   # 192, 192, 96, 96
    per_group = gpg * P * 4 * (widths[0] + widths[1] + widths[3])     # h1, h2 live together; h3 reuses h1's buffer
    chunk = max(1, min(n_groups, PAIR_MLP_BYTES // max(per_group, 1)))
    tiles_m = int(lib.mft_pair_mlp_tiles_m(gpg, N))
    rows_c = chunk * gpg * P
    # raw layer outputs of one chunk, shared by all Wcompute instances of the head (same tag-independent names)
    hbuf = [arena.get("pm.h%d" % i, (rows_c * w,)) for i, w in ((0, widths[0]), (1, widths[1]), (3, widths[3]))]
    ws_mean = arena.get("pm.wsm", (chunk * tiles_m * 192,))
    ws_m2 = arena.get("pm.wsq", (chunk * tiles_m * 192,))
    ws_n = arena.get("pm.wsn", (chunk * tiles_m,))
    scale = [arena.get("pm.sc%d" % i, (chunk, w)) for i, w in enumerate(widths)]
    shift = [arena.get("pm.sh%d" % i, (chunk, w)) for i, w in enumerate(widths)]
    s_ut = arena.get("pm.s", (rows_c,))
    Kp = ops.round_up(F, 32)
    ld = x.shape[1]
    st = ops._stream
    for g0 in range(0, n_groups, chunk):
        ng = min(chunk, n_groups - g0)
        xin = x[g0 * gpg * N:]
        outs = [hbuf[0], hbuf[1], hbuf[0], hbuf[2]]                   # h1 -> h2 -> h3 (over h1) -> h4
        h_in, ld_in, K, Kpad = xin, ld, F, Kp
        for li, (w, b, gam, beta, cout) in enumerate(layers):
            ops._lib.check(lib.mft_pair_mlp_layer(ops._p(h_in), ld_in, 0 if li == 0 else 1, ops._p(ij),
                                                  ops._p(scale[li - 1]) if li else None, ops._p(shift[li - 1]) if li else None,
                                                  ops._p(w), K, Kpad, ops._p(b), ops._p(outs[li]), cout, ng, gpg, N,
                                                  ops.LRELU_SLOPE, ops._p(ws_mean), ops._p(ws_m2), ops._p(ws_n), 1 if PAIR_F16X2 else 0, st()),
                           "mft_pair_mlp_layer")
            ops._lib.check(lib.mft_pair_mlp_stats_finalize(ops._p(ws_mean), ops._p(ws_m2), ops._p(ws_n), ng, tiles_m, cout,
                                                           ops._p(gam), ops._p(beta), ops.BN_EPS, ops._p(scale[li]),
                                                           ops._p(shift[li]), None, None, st()), "mft_pair_mlp_stats_finalize")
            h_in, ld_in, K, Kpad = outs[li], cout, cout, cout
        ops._lib.check(lib.mft_pair_mlp_score(ops._p(h_in), widths[3], ops._p(scale[3]), ops._p(shift[3]), ops._p(w5), ops._p(b5),
                                              ops.LRELU_SLOPE, ops._p(s_ut), ng, gpg, N, st()), "mft_pair_mlp_score")
        ops._lib.check(lib.mft_masked_softmax_ut(ops._p(s_ut), ops._p(A[g0 * gpg:]), ng * gpg, N, st()), "mft_masked_softmax_ut")
    return A


def wcompute_unfused(G, name, x, F, n_graphs, N, n_groups, arena, tag="wc"):
    """Round-1 form of ``wcompute``: the materialised pair tensor through generic GEMM / BatchNorm launches (kept as the A/B and
    test reference of the fused kernels; MFT_FUSED_PAIR_MLP=0)."""
    layers, (w5, b5) = G.wc[name]
    Kp = ops.round_up(F, 32)
    rows = n_graphs * N * N
    rpg = rows // n_groups
    lib = ops._lib.lib()
    d = arena.get(tag + ".d", (rows, Kp))
    ops._lib.check(lib.mft_pair_absdiff(ops._p(x), x.shape[1], ops._p(d), Kp, n_graphs, N, F, ops._stream()),
                   "mft_pair_absdiff")
    h, K = d, Kp
    for li, (w, b, g, beta, cout) in enumerate(layers):
        o = ops.gemm(h, K, w, cout, bias=b, out=arena.get(tag + ".h%d" % li, (rows, cout)))
        m, s = _bn_stats4(arena, tag + ".bn%d" % li, o.view(rows, 1, 1, cout), rpg, n_groups)
        ops.bn_apply(o, cout, rpg, n_groups, m, s, g, beta, act=ops.ACT_LRELU, out=o)
        h, K = o, cout
    sc = ops.gemm(h, K, w5, 1, bias=b5, out=arena.get(tag + ".s", (rows, 1)))
    A = arena.get(tag + ".A", (n_graphs, N, N))
    ops._lib.check(lib.mft_masked_softmax(ops._p(sc), 1, ops._p(A), n_graphs, N, ops._stream()), "mft_masked_softmax")
    return A


def gconv(G, name, A, x, F, n_graphs, N, n_groups, arena, tag="gc"):
    """gnn.Gconv.forward with gmul (gnn.py:16-56): fc(cat(x, A@x)) [+ BatchNorm1d over the group's graphs*N rows]."""
    w, b, g, beta, cout = G.gc[name]
    rows = n_graphs * N
    ldy = ops.round_up(2 * F, 32)
    y = arena.get(tag + ".y", (rows, ldy))
    ops._lib.check(ops._lib.lib().mft_graph_aggregate(ops._p(A), ops._p(x), x.shape[1], ops._p(y), ldy, n_graphs, N, F,
                                                      ops._stream()), "mft_graph_aggregate")
    o = ops.gemm(y, ldy, w, cout, bias=b, out=arena.get(tag + ".o", (rows, cout)))
    if g is not None:
        rpg = rows // n_groups
        m, s = _bn_stats4(arena, tag + ".bn", o.view(rows, 1, 1, cout), rpg, n_groups)
        ops.bn_apply(o, cout, rpg, n_groups, m, s, g, beta, act=ops.ACT_NONE, out=o)
    return o


def gnn_forward(G, nodes, n_graphs, N, n_groups, arena, tag="gnn"):
    """gnn.GNN_nl.forward (gnn.py:154-166).  nodes [n_graphs*N, ld>=256] with features in columns 0..132+.
    Returns [n_graphs*N, n_way].  ``nodes`` is extended in place (x = cat(x, x_new))."""
    F = 128 + G.n_way
    x = nodes
    for i in range(2):
        A = wcompute(G, "layer_w%d" % i, x, F, n_graphs, N, n_groups, arena, tag + ".w%d" % i)
        o = gconv(G, "layer_l%d" % i, A, x, F, n_graphs, N, n_groups, arena, tag + ".l%d" % i)
        ops.copy_cols(o, x, F, o.shape[1], act=ops.ACT_LRELU)
        F += o.shape[1]
    A = wcompute(G, "w_comp_last", x, F, n_graphs, N, n_groups, arena, tag + ".wl")
    return gconv(G, "layer_last", A, x, F, n_graphs, N, n_groups, arena, tag + ".ll")


def gnnnet_scores(G, feats, n_episodes, n_way, n_support, n_query, arena, fold=False, tag="head"):
    """GnnNet.set_forward(is_feature=True) tail (gnnnet.py:71-87,210-217) for ``n_episodes`` episodes at once:
    feats [n_episodes*n_way*(S+n_query), 512] -> scores [n_episodes*n_way*n_query, n_way]."""
    rows = feats.shape[0]
    rpg = rows // n_episodes
    z = ops.gemm(feats, 512, G.fc_w, 128, bias=G.fc_b, out=arena.get(tag + ".z", (rows, 128)))
    m, s = _bn_stats4(arena, tag + ".fcbn", z.view(rows, 1, 1, 128), rpg, n_episodes)
    ops.bn_apply(z, 128, rpg, n_episodes, m, s, G.fc_g, G.fc_beta, act=ops.ACT_NONE, out=z)
    N = n_way * (n_support + 1)
    n_graphs = n_episodes * n_query
    nodes = arena.get(tag + ".nodes", (n_graphs * N, 256))
    ops._lib.check(ops._lib.lib().mft_build_graph_nodes(ops._p(z), 128, ops._p(nodes), 256, n_episodes, n_way,
                                                        n_support, n_query, 1 if fold else 0, ops._stream()),
                   "mft_build_graph_nodes")
    out = gnn_forward(G, nodes, n_graphs, N, n_episodes, arena, tag + ".gnn")
    return ops.gather_query_scores(out, n_episodes, n_way, n_support, n_query)
