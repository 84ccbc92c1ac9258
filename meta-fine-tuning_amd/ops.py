"""Tensor-level wrappers over the C-ABI kernels (include/mft_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the
current HIP stream; every arithmetic op is a launch into libmft_hip.so.  All
activations are NHWC fp32 CUDA tensors.  Nothing in this module falls back to
torch math: a CPU tensor or a missing library raises.
"""
import ctypes
import os

import torch

from . import _lib
from . import settings

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
BN_EPS = 1e-5
LRELU_SLOPE = 0.01


import threading

_tls = threading.local()      # per host thread: the device of the pointer arguments marshalled for the launch being assembled


def _stream(t=None):
    """The current HIP stream of the device THIS launch's tensors live on.  ``t``: a tensor / torch.device / index names the
    device explicitly; without it the device is the one the ``_p()`` calls of the same argument list saw (Python evaluates
    arguments left to right and every launcher takes the stream last).  The record is per host thread and is consumed here, so
    it can neither leak into a later launch nor be overwritten by another thread's engine on another device; a launch whose
    pointer arguments sit on two devices raises."""
    dev = getattr(_tls, "dev", None)
    _tls.dev = None
    if t is not None:
        dev = t.device.index if torch.is_tensor(t) else (t.index if isinstance(t, torch.device) else int(t))
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("meta_fine_tuning_amd ops need CUDA (HIP) tensors; got a CPU tensor (no CPU fallback)")
    prev = getattr(_tls, "dev", None)
    if prev is not None and prev != t.device.index and getattr(_tls, "strict", True):
        _tls.dev = None
        raise RuntimeError("one launch got pointers on cuda:%d and cuda:%d" % (prev, t.device.index))
    _tls.dev = t.device.index
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("expected a contiguous float32 tensor, got %s contiguous=%s" % (t.dtype, t.is_contiguous()))
    return t


def round_up(a, b):
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------ layout

def nchw_to_nhwc(x):
    """[n,C,H,W] -> [n,H,W,C] (boundary ingest)."""
    _f32c(x)
    n, C, H, W = x.shape
    y = torch.empty((n, H, W, C), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_nchw_to_nhwc(_p(x), _p(y), n, C, H, W, _stream()), "mft_nchw_to_nhwc")
    return y


def pack_conv_weight(w, rows32=False):
    """OIHW (or [out,in] Linear) -> packed [Cout, roundup(KH*KW*Cin,32)].  ``rows32``: the pack lives in a buffer whose row count is
    rounded up to 32 with zero rows (``pk.rows32``: that buffer) -- the operand the data-gradient launch wants when Cout % 32 != 0
    (the reduction runs over Cout); a refresh through PackPlan rewrites the Cout real rows only, the padding stays zero."""
    if w.dim() == 2:
        w = w.view(w.shape[0], w.shape[1], 1, 1)
    w = _f32c(w.detach())
    Cout, Cin, KH, KW = w.shape
    kpad = round_up(KH * KW * Cin, 32)
    if rows32 and Cout % 32 != 0:
        buf = torch.zeros((round_up(Cout, 32), kpad), device=w.device, dtype=torch.float32)
        pk = buf[:Cout]
        pk.rows32 = buf
    else:
        pk = torch.empty((Cout, kpad), device=w.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_pack_oihw(_p(w), _p(pk), Cout, Cin, KH, KW, kpad, _stream()), "mft_pack_oihw")
    return pk


class PackPlan:
    """The OIHW -> packed copies of a set of live parameters as ONE launch (mft_pack_oihw_multi): ``add`` registers
    (source parameter, packed destination) pairs -- both must stay where they are --, ``run`` refreshes every destination."""

    def __init__(self):
        self.jobs, self.total, self.table, self.keep = [], 0, None, []

    def add(self, w, pk):
        w4 = w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)
        Cout, Cin, KH, KW = w4.shape
        assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and pk.is_contiguous() and pk.shape == (Cout, round_up(KH * KW * Cin, 32))
        self.jobs.append((w.data_ptr(), pk.data_ptr(), Cout, Cin, KH * KW, pk.shape[1], self.total))
        self.total += pk.numel()
        self.keep.append((w, pk))
        self.table = None

    def add_transposed(self, w, pkT):
        """``w`` [Cout, Cin] (or [Cout, Cin, 1, 1]) -> ``pkT`` [rows >= Cin, roundup(Cout, 32)] = W^T zero-padded: the operand with which
        the FORWARD GEMM kernel computes a linear layer's data gradient dx = dy @ W (10-27 % faster per launch than the BT form reading
        the forward pack, bit-identical: tools/gemm_vs_dgrad_1x1.py), refreshed by the same launch as the forward packs."""
        Cout, Cin = w.shape[0], w.shape[1]
        assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.numel() == Cout * Cin
        assert pkT.is_contiguous() and pkT.shape[0] >= Cin and pkT.shape[1] == round_up(Cout, 32)
        self.jobs.append((w.data_ptr(), pkT.data_ptr(), Cout, Cin, -1, pkT.shape[1], self.total))
        self.total += pkT.numel()
        self.keep.append((w, pkT))
        self.table = None

    def sources(self):
        return [w for w, _ in self.keep]

    def run(self):
        if not self.jobs:
            return
        if self.table is None:
            import numpy as np
            self.table = torch.from_numpy(np.asarray(self.jobs, dtype=np.int64)).to(self.keep[0][1].device)
        _lib.check(_lib.lib().mft_pack_oihw_multi(_p(self.table), len(self.jobs), self.total, _stream()), "mft_pack_oihw_multi")


class SplitPlan:
    """bf16x3 planes of several packed weight matrices refreshed by ONE launch (mft_split_bf16x3_multi): the per-step companion of
    PackPlan for the meta-training layers that run on the split-precision kernels.  ``add(pk, Cout, Cin, taps, transposed)``
    allocates the planes [3, rows, K] (rows, K = Cout, taps*Cin -- or Cin, taps*Cout for the transposed data-gradient operand),
    fills them once and registers the job; ``run()`` refreshes every registered set from its (in-place updated) source."""

    def __init__(self):
        self.jobs, self.total, self.table, self.keep = [], 0, None, []
        self.retired = []          # job tables an earlier recording may still read: kept alive for good

    def add(self, pk, Cout, Cin, taps, transposed=False):
        assert pk.is_cuda and pk.dtype == torch.float32 and pk.is_contiguous() and pk.shape == (Cout, taps * Cin) and (taps * Cin) % 32 == 0
        if torch.cuda.is_current_stream_capturing():
            # a new job uploads a table (a synchronous host-to-device copy) and fills planes outside any recording: the episode
            # loops run three eager steps before they record, so every layer's job exists by then
            raise RuntimeError("SplitPlan.add() under stream capture: run one eager step first (the planes of every layer are "
                               "registered by the first forward)")
        n = pk.numel()
        planes = torch.empty((3, Cin, taps * Cout) if transposed else (3, Cout, taps * Cin), device=pk.device, dtype=torch.int16)
        self.jobs.append((pk.data_ptr(), planes.data_ptr(), n, Cout, Cin, taps, 1 if transposed else 0, self.total))
        self.total += n
        self.keep.append((pk, planes))
        if self.table is not None:
            self.retired.append(self.table)      # a hipGraph recorded before this job existed replays run() with the OLD table: it must
        self.table = None                        # stay valid (it refreshes the jobs that recording uses; the next run() builds the new one)
        self._launch(len(self.jobs) - 1)
        return planes

    def _launch(self, only=None):
        import numpy as np
        if only is not None:                       # a job that has just been added: fill its planes now, alone
            j = list(self.jobs[only])
            j[7] = 0
            tab = torch.from_numpy(np.asarray([j], dtype=np.int64)).to(self.keep[only][0].device)
            _lib.check(_lib.lib().mft_split_bf16x3_multi(_p(tab), 1, j[2], _stream()), "mft_split_bf16x3_multi")
            return
        if self.table is None:
            self.table = torch.from_numpy(np.asarray(self.jobs, dtype=np.int64)).to(self.keep[0][0].device)
        _lib.check(_lib.lib().mft_split_bf16x3_multi(_p(self.table), len(self.jobs), self.total, _stream()), "mft_split_bf16x3_multi")

    def run(self):
        if self.jobs:
            self._launch()


def unpack_conv_weight(pk, shape):
    Cout, Cin, KH, KW = shape
    w = torch.empty(shape, device=pk.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_unpack_oihw(_p(pk), _p(w), Cout, Cin, KH, KW, pk.shape[-1], _stream()), "mft_unpack_oihw")
    return w


def pack_dgrad_weight(w_pk, Cout, Cin, KH, KW, groups=1):
    """packed fwd weights [groups, Cout, KH*KW*Cin] -> dgrad weights [groups, Cin, KH*KW*Cout]."""
    w_pk = _f32c(w_pk)
    K = KH * KW * Cin
    wt = torch.empty((groups, Cin, KH * KW * Cout), device=w_pk.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_pack_dgrad(_p(w_pk), _p(wt), Cout, Cin, KH, KW, groups, Cout * K, Cin * KH * KW * Cout,
                                         _stream()), "mft_pack_dgrad")
    return wt


# ------------------------------------------------------------------------------------ conv / gemm

def conv2d(x, w_pk, Cout, KH, KW, stride, pad, imgs_per_group=0, bias=None, out=None):
    """x [n,H,W,Cin] NHWC, w_pk [Cout,Kpad] or [groups,Cout,Kpad] -> [n,OH,OW,Cout]."""
    _f32c(x)
    n, H, W, Cin = x.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((n, OH, OW, Cout), device=x.device, dtype=torch.float32)
    wgs = 0
    if w_pk.dim() == 3:
        wgs = w_pk.shape[1] * w_pk.shape[2]
        if imgs_per_group <= 0 or n // imgs_per_group != w_pk.shape[0]:
            raise RuntimeError("per-group weights need imgs_per_group with n/imgs_per_group == groups")
    if ((wgs == 0 and imgs_per_group <= 0) or (w_pk.dim() == 3 and w_pk.shape[0] == 1)) and Cin != 3:        # (one group = one weight set)
        # one weight set, few output tiles, long reduction (one 105-image meta-training episode): K sliced across workgroups
        nws = _ksplit_ws(n * OH * OW, Cout, w_pk.shape[-1])
        if nws:
            ws = torch.empty((nws,), device=x.device, dtype=torch.float32)
            rc = _lib.lib().mft_conv2d_nhwc_ksplit(_p(x), Cin, _p(w_pk), _p(bias), _p(out), Cout, n, H, W, Cin, Cout, KH, KW,
                                                   stride, pad, _p(ws), _stream())
            _lib.check(rc, "mft_conv2d_nhwc_ksplit")
            return out
    if w_pk.dim() == 3 and 1 < w_pk.shape[0] <= SMALL_GROUPS and bias is None and Cin != 3:
        # a few episodes in lockstep: grid.y = episode on the K-sliced implicit GEMM (the weight-streaming kernels need >= 16
        # episodes to fill the GPU)
        G = w_pk.shape[0]
        nws = _ksplit_ws(imgs_per_group * OH * OW, Cout, w_pk.shape[-1], G)
        if nws:
            ws = torch.empty((nws,), device=x.device, dtype=torch.float32)
            rc = _lib.lib().mft_conv2d_nhwc_ksplit_grouped(_p(x), Cin, _p(w_pk), _p(out), Cout, n, H, W, Cin, Cout, KH, KW, stride, pad,
                                                           imgs_per_group, wgs, _p(ws), _stream())
            _lib.check(rc, "mft_conv2d_nhwc_ksplit_grouped")
            return out
    rc = _lib.lib().mft_conv2d_nhwc(_p(x), Cin, _p(w_pk), _p(bias), _p(out), Cout, n, H, W, Cin, Cout, KH, KW,
                                    stride, pad, imgs_per_group, wgs, _stream())
    _lib.check(rc, "mft_conv2d_nhwc")
    return out


SMALL_GROUPS = settings.current().small_groups     # up to this many per-episode weight sets take the K-sliced GEMM route
_KSPLIT_WS = {}


def _ksplit_ws(rows, cols, K, groups=1):
    """floats of workspace for the K-sliced convolution / data-gradient launch of this shape (0: not sliced); memoised."""
    key = (rows, cols, K, groups)
    v = _KSPLIT_WS.get(key)
    if v is None:
        v = _KSPLIT_WS[key] = int(_lib.lib().mft_conv_ksplit_ws_floats(rows, cols, K) if groups == 1 else
                                  _lib.lib().mft_conv_ksplit_grouped_ws_floats(rows, cols, K, groups))
    return v


def split_weight_x3(w_pk):
    """packed fp32 weights [Cout, Kpad] -> three bf16 planes [3, Cout, Kpad] (int16 storage) for conv2d_x3."""
    _f32c(w_pk)
    planes = torch.empty((3,) + tuple(w_pk.shape), device=w_pk.device, dtype=torch.int16)
    _lib.check(_lib.lib().mft_split_bf16x3(_p(w_pk), _p(planes), w_pk.numel(), _stream()), "mft_split_bf16x3")
    return planes


def split_weight_h2(w_pk):
    """packed fp32 weights [Cout, Kpad] -> two fp16 planes [2, Cout, Kpad] (int16 storage): hi = fp16(w), lo = fp16((w - hi) * 2^11)
    (csrc/conv_x3.hip, "f16x2").  The conv2d_x3* wrappers below take either kind of planes and launch the matching kernels."""
    _f32c(w_pk)
    planes = torch.empty((2,) + tuple(w_pk.shape), device=w_pk.device, dtype=torch.int16)
    _lib.check(_lib.lib().mft_split_f16x2(_p(w_pk), _p(planes), w_pk.numel(), _stream()), "mft_split_f16x2")
    return planes


def _x3_fn(w3, stem):
    """The C entry point for these weight planes: three bf16 planes -> mft_conv2d_nhwc_x3*, two fp16 planes -> mft_conv2d_nhwc_h2*."""
    kind = {3: "x3", 2: "h2"}[int(w3.shape[0])]
    name = "mft_conv2d_nhwc_%s%s" % (kind, stem)
    return getattr(_lib.lib(), name), name


def conv2d_x3(x, w3, Cout, KH, KW, stride, pad, out=None):
    """conv2d for frozen shared weights on the bf16 / fp16 matrix cores with fp32 accuracy (6-term bf16x3 or 3-term f16x2 products,
    by the kind of ``w3``).  x [n,H,W,Cin] fp32, w3 = split_weight_x3 / split_weight_h2(pack_conv_weight(w)) -> [n,OH,OW,Cout] fp32."""
    _f32c(x)
    n, H, W, Cin = x.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((n, OH, OW, Cout), device=x.device, dtype=torch.float32)
    fn, name = _x3_fn(w3, "")
    rc = fn(_p(x), Cin, _p(w3), w3.shape[1] * w3.shape[2], _p(out), Cout, n, H, W, Cin, Cout, KH, KW, stride, pad, _stream())
    _lib.check(rc, name)
    return out


def conv2d_x3_bnstats(x, w3, Cout, KH, KW, stride, pad, imgs_per_group, out, ws, mean, rstd, eps=BN_EPS):
    """conv2d_x3 that also returns the following train-mode BatchNorm's statistics per group of ``imgs_per_group`` images
    (computed from the output tiles while they are still in registers).  ``ws``: >= mft_conv2d_x3_stats_ws_floats floats."""
    _f32c(x)
    n, H, W, Cin = x.shape
    fn, name = _x3_fn(w3, "_bnstats")
    rc = fn(_p(x), Cin, _p(w3), w3.shape[1] * w3.shape[2], _p(out), Cout, n, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, eps,
            _p(ws), _p(mean), _p(rstd), _stream())
    if rc == _lib.MFT_EINVAL:
        return None                       # outside the fused form's domain (e.g. an input of 2 GiB or more): caller uses conv + bn_stats
    _lib.check(rc, name)
    return out, mean, rstd


def conv2d_x3_bnin_bnstats(c_raw, in_ws, in_gamma, in_beta, w3, Cout, imgs_per_group, out, ws, mean=None, rstd=None, eps=BN_EPS):
    """3x3 / stride 1 / pad 1 conv2d_x3 of relu(BatchNorm(c_raw)): ``c_raw`` [n,H,W,Cin] is the raw output of the previous
    bf16x3 convolution, ``in_ws`` its statistics partials (SimpleBlock's C1 -> BN1 -> ReLU -> C2, backbone.py:251-256, as one
    launch).  This convolution's own partials go to ``ws``; ``mean`` / ``rstd`` None = leave them as partials for the consumer.
    Returns None outside the kernel's domain."""
    _f32c(c_raw)
    n, H, W, Cin = c_raw.shape
    fn, name = _x3_fn(w3, "_bnin_bnstats")
    rc = fn(_p(c_raw), Cin, _p(in_ws), _p(in_gamma), _p(in_beta), _p(w3), w3.shape[1] * w3.shape[2], _p(out), Cout, n, H, W, Cin, Cout,
            imgs_per_group, eps, _p(ws), _p(mean), _p(rstd), _stream())
    if rc == _lib.MFT_EINVAL:
        return None
    _lib.check(rc, name)
    return out


def bn_apply_x3ws(x2d, C, rows_per_group, n_groups, ws, gamma, beta, out, act=ACT_NONE, res=None, res_ws=None, res_gamma=None,
                  res_beta=None, slope=LRELU_SLOPE, eps=BN_EPS, stats=None, res_stats=None):
    """bn_apply for the output of a bf16x3 convolution whose statistics are still per-tile partials ``ws`` (and ``res_ws`` for a
    BatchNorm'd residual branch): the merge happens in the prologue of the apply launch.  ``stats`` / ``res_stats`` = optional
    (mean, rstd) [n_groups, C] buffers that receive the merged statistics."""
    _f32c(x2d)
    m, r = stats if stats is not None else (None, None)
    rm, rr = res_stats if res_stats is not None else (None, None)
    rc = _lib.lib().mft_bn_apply_x3ws(_p(x2d), x2d.shape[-1], _p(out), out.shape[-1], C, rows_per_group, n_groups, _p(ws), _p(gamma),
                                      _p(beta), _p(res), 0 if res is None else res.shape[-1], _p(res_ws), _p(res_gamma),
                                      _p(res_beta), act, slope, eps, _p(m), _p(r), _p(rm), _p(rr), _stream())
    _lib.check(rc, "mft_bn_apply_x3ws")
    return out


def conv2d_x3p_bnstats(xp, n, H, W, w3, Cout, KH, KW, stride, pad, imgs_per_group, out, ws, mean, rstd, eps=BN_EPS):
    """conv2d_x3_bnstats on a pre-split input: ``xp`` int16 [3, n*H*W, Cin] (bf16x3 planes written by bn_apply(planes=) /
    bn_relu_maxpool_gather(planes=)); the operand path of the convolution is then a plain copy."""
    Cin = xp.shape[2]
    if w3.shape[0] != 3:
        raise ValueError("pre-split activation planes exist for the bf16x3 kernels only")
    rc = _lib.lib().mft_conv2d_nhwc_x3p_bnstats(_p(xp), xp.shape[1] * Cin, Cin, _p(w3), w3.shape[1] * w3.shape[2], _p(out), Cout,
                                                n, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, eps, _p(ws), _p(mean),
                                                _p(rstd), _stream())
    _lib.check(rc, "mft_conv2d_nhwc_x3p_bnstats")
    return out, mean, rstd


def gemm(a, K, w_pk, N, bias=None, out=None, ldo=None, rows_per_group=0):
    """out[m, :N] = a[m, :K] @ w_pk[:N, :K].T + bias.  `a` is [M, lda] with lda >= K, K % 32 == 0."""
    _f32c(a)
    M, lda = a.shape
    if out is None:
        ldo = N if ldo is None else ldo
        out = torch.empty((M, ldo), device=a.device, dtype=torch.float32)
    else:
        ldo = out.shape[1]
    wgs = 0
    if w_pk.dim() == 3:
        wgs = w_pk.shape[1] * w_pk.shape[2]
    rc = _lib.lib().mft_conv2d_nhwc(_p(a), lda, _p(w_pk), _p(bias), _p(out), ldo, M, 1, 1, K, N, 1, 1, 1, 0,
                                    rows_per_group, wgs, _stream())
    _lib.check(rc, "mft_conv2d_nhwc(gemm)")
    return out


def gemm_rk(a, K, w_pk, N, bias=None, out=None, ldo=None):
    """``gemm`` for the skinny linear layers of a meta-training step (a few hundred rows, K <= 512): the register-K kernel
    (mft_gemm_rk, csrc/gnn.hip) -- 16-row tiles, no LDS staging.  Same contract as ``gemm`` for a 2-D pack; columns N.. of ``out``
    are left alone."""
    _f32c(a)
    M, lda = a.shape
    assert w_pk.dim() == 2 and w_pk.shape[1] == K and w_pk.shape[0] >= N and K % 16 == 0 and K <= 512, (tuple(w_pk.shape), K, N)
    if out is None:
        ldo = N if ldo is None else ldo
        out = torch.empty((M, ldo), device=a.device, dtype=torch.float32)
    else:
        ldo = out.shape[1]
    _lib.check(_lib.lib().mft_gemm_rk(_p(a), lda, _p(w_pk), w_pk.shape[0], K, _p(bias), _p(out), ldo, M, N, _stream()), "mft_gemm_rk")
    return out


def conv2d_dgrad(dy, w_pk, Cin, KH, KW, pad, imgs_per_group=0, out=None, stride=1, in_hw=None):
    """dy [n,OH,OW,Cout], forward weight pack [Cout,KH*KW*Cin] or [groups,...] -> dx [n,H,W,Cin].
    ``in_hw`` = (H, W) of the forward input (needed when stride > 1; defaults to dy's size for stride 1)."""
    _f32c(dy)
    n, OH, OW, Cout = dy.shape
    H, W = (OH, OW) if in_hw is None else in_hw
    if out is None:
        out = torch.empty((n, H, W, Cin), device=dy.device, dtype=torch.float32)
    wgs = w_pk.shape[1] * w_pk.shape[2] if w_pk.dim() == 3 else 0
    if ((wgs == 0 and imgs_per_group <= 0) or (w_pk.dim() == 3 and w_pk.shape[0] == 1)) and Cin % 64 == 0:
        nws = _ksplit_ws(n * H * W, Cin, KH * KW * Cout)
        if nws:
            ws = torch.empty((nws,), device=dy.device, dtype=torch.float32)
            rc = _lib.lib().mft_conv2d_dgrad_nhwc_ksplit(_p(dy), Cout, _p(w_pk), _p(out), Cin, n, H, W, Cin, Cout, KH, KW, stride, pad,
                                                         _p(ws), _stream())
            _lib.check(rc, "mft_conv2d_dgrad_nhwc_ksplit")
            return out
    if w_pk.dim() == 3 and 1 < w_pk.shape[0] <= SMALL_GROUPS and Cin % 64 == 0 and stride == 1 and imgs_per_group > 0:
        nws = _ksplit_ws(imgs_per_group * H * W, Cin, KH * KW * Cout, w_pk.shape[0])
        if nws:
            ws = torch.empty((nws,), device=dy.device, dtype=torch.float32)
            rc = _lib.lib().mft_conv2d_dgrad_nhwc_ksplit_grouped(_p(dy), Cout, _p(w_pk), _p(out), Cin, n, H, W, Cin, Cout, KH, KW, stride,
                                                                 pad, imgs_per_group, wgs, _p(ws), _stream())
            _lib.check(rc, "mft_conv2d_dgrad_nhwc_ksplit_grouped")
            return out
    rc = _lib.lib().mft_conv2d_dgrad_nhwc(_p(dy), Cout, _p(w_pk), _p(out), Cin, n, H, W, Cin, Cout, KH, KW, stride, pad,
                                          imgs_per_group, wgs, _stream())
    _lib.check(rc, "mft_conv2d_dgrad_nhwc")
    return out


def conv2d_wgrad(x, dy, Cout, KH, KW, stride, pad, imgs_per_group=0, out=None, ldy=None):
    """x [n,H,W,Cin], dy [n,OH,OW,Cout] (row stride ldy) -> dw [groups, Cout, roundup(KH*KW*Cin,32)] (packed layout)."""
    _f32c(x)
    _f32c(dy)
    n, H, W, Cin = x.shape
    groups = 1 if imgs_per_group <= 0 else n // imgs_per_group
    K = round_up(KH * KW * Cin, 32)
    if out is None:
        out = torch.empty((groups, Cout, K), device=x.device, dtype=torch.float32)
    nws = int(_lib.lib().mft_conv2d_wgrad_ws_floats(n, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group))
    ws = torch.empty((nws,), device=x.device, dtype=torch.float32) if nws > 0 else None
    rc = _lib.lib().mft_conv2d_wgrad_nhwc(_p(x), Cin, _p(dy), dy.shape[-1] if ldy is None else ldy, _p(out), n, H, W,
                                          Cin, Cout, KH, KW, stride, pad, imgs_per_group, Cout * K, _p(ws), _stream())
    _lib.check(rc, "mft_conv2d_wgrad_nhwc")
    return out


def conv2d_wgrad_oihw(x, dy, Cout, KH, KW, stride, pad, cin_valid=0, cout_valid=0, out=None):
    """Weight gradient of a shared-weight convolution as torch lays Conv2d.weight.grad out: [Cout, Cin, KH, KW].
    ``cin_valid`` / ``cout_valid``: the operands carry zero-padded channels; only that corner is written (contiguously)."""
    _f32c(x)
    _f32c(dy)
    n, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((cout_valid or Cout, cin_valid or Cin, KH, KW), device=x.device, dtype=torch.float32)
    ws = torch.empty((int(_lib.lib().mft_conv2d_wgrad_oihw_ws_floats(n, H, W, Cin, Cout, KH, KW, stride, pad)),), device=x.device,
                     dtype=torch.float32)
    rc = _lib.lib().mft_conv2d_wgrad_oihw(_p(x), Cin, _p(dy), dy.shape[-1], _p(out), n, H, W, Cin, Cout, KH, KW, stride, pad,
                                          cin_valid, cout_valid, _p(ws), _stream())
    _lib.check(rc, "mft_conv2d_wgrad_oihw")
    return out


class _WgradJob(ctypes.Structure):
    """MftWgradJob (include/mft_hip.h)"""
    _fields_ = [("inp", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("ws", ctypes.c_void_p)] + \
               [(n, ctypes.c_int) for n in ("ldi", "ldy", "n_img", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "cin_valid",
                                            "cout_valid", "reserved")]


class WgradBatch:
    """Weight gradients of one backward pass, DEFERRED: ``add`` takes the same arguments as ``conv2d_wgrad_oihw`` and returns the
    output tensor at once -- its contents exist after ``flush()``, which runs all registered problems in ONE pair of launches per
    16 jobs (mft_conv2d_wgrad_oihw_multi).  Nothing downstream of a layer reads its weight gradient, so the meta-training backward
    registers every layer's as it goes and flushes at the end: 31 + 31 launches per step become 2 + 2 (+ the stem's own), each job
    bit-identical to its own launch.  The operands are kept alive (and must not be overwritten) until the flush.
    ``MFT_WGRAD_BATCH=0``: every ``add`` launches immediately (A/B and bit-identity reference)."""

    def __init__(self, enabled=True):
        self.enabled = enabled
        self.jobs, self.keep, self.after = [], [], []

    def add(self, x, dy, Cout, KH, KW, stride, pad, cin_valid=0, cout_valid=0):
        if not self.enabled:
            return conv2d_wgrad_oihw(x, dy, Cout, KH, KW, stride, pad, cin_valid, cout_valid)
        _f32c(x)
        _f32c(dy)
        n, H, W, Cin = x.shape
        out = torch.empty((cout_valid or Cout, cin_valid or Cin, KH, KW), device=x.device, dtype=torch.float32)
        ws = torch.empty((int(_lib.lib().mft_conv2d_wgrad_oihw_ws_floats(n, H, W, Cin, Cout, KH, KW, stride, pad)),), device=x.device,
                         dtype=torch.float32)
        self.jobs.append((x.data_ptr(), dy.data_ptr(), out.data_ptr(), ws.data_ptr(), Cin, dy.shape[-1], n, H, W, Cin, Cout, KH, KW, stride,
                          pad, cin_valid, cout_valid, 0))
        self.keep.append((x, dy, out, ws))
        return out

    def then(self, fn):
        """Run ``fn()`` right after the flush (work that reads a deferred gradient, e.g. the sum over row chunks)."""
        if not self.enabled:
            fn()
        else:
            self.after.append(fn)

    def flush(self):
        if self.jobs:
            arr = (_WgradJob * len(self.jobs))(*[_WgradJob(*j) for j in self.jobs])
            dev = self.keep[0][0].device
            _lib.check(_lib.lib().mft_conv2d_wgrad_oihw_multi(arr, len(self.jobs), _stream(dev)), "mft_conv2d_wgrad_oihw_multi")
        for fn in self.after:
            fn()
        self.jobs, self.keep, self.after = [], [], []


def conv2d_wgrad_adam(x, dy, w, m, v, Cout, KH, KW, stride, pad, step, imgs_per_group=0, lr=0.01, beta1=0.9,
                      beta2=0.999, eps=1e-8, dw=None, hyper=None):
    """Weight gradient of a conv with the Adam update of (w, m, v) [groups, Cout, KH*KW*Cin] fused in the epilogue.
    ``hyper``: device tensor {lr/(1-b1^t), 1/sqrt(1-b2^t)} maintained by adam_hyper_advance (hipGraph replay); replaces step/lr."""
    _f32c(x)
    _f32c(dy)
    n, H, W, Cin = x.shape
    K = KH * KW * Cin
    if hyper is not None:
        rc = _lib.lib().mft_conv2d_wgrad_adam_nhwc_dev(_p(x), Cin, _p(dy), Cout, _p(w), _p(m), _p(v), _p(dw), n, H, W, Cin,
                                                       Cout, KH, KW, stride, pad, imgs_per_group, Cout * K, _p(hyper), beta1,
                                                       beta2, eps, _stream())
        _lib.check(rc, "mft_conv2d_wgrad_adam_nhwc_dev")
        return
    rc = _lib.lib().mft_conv2d_wgrad_adam_nhwc(_p(x), Cin, _p(dy), Cout, _p(w), _p(m), _p(v), _p(dw), n, H, W, Cin,
                                               Cout, KH, KW, stride, pad, imgs_per_group, Cout * K, step, lr, beta1,
                                               beta2, eps, _stream())
    _lib.check(rc, "mft_conv2d_wgrad_adam_nhwc")


WF_RAW, WF_ENTRY, WF_EXIT = 0, 1, 2


def wgrad_adam_next_forward(x, dy, w, m, v, KH, KW, stride, pad, step, imgs_per_group, x_next=None, mode=WF_RAW, raw=None, act=None,
                            gamma=None, beta=None, gbs=0, mean=None, rstd=None, sc_raw=None, gamma_s=None, beta_s=None, mean_s=None,
                            rstd_s=None, pooled=None, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, dw=None, hyper=None):
    """Weight gradient + Adam of (w, m, v) [groups, Cout, KH*KW*Cin] for inner step t and -- with ``x_next`` -- the same layer's
    convolution of step t+1 from the weight tiles just updated (csrc/wgrad_fwd.hip).  Returns False outside the kernel's domain."""
    _f32c(x)
    _f32c(dy)
    n, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    rc = _lib.lib().mft_wgrad_adam_next_forward(
        _p(x), Cin, _p(dy), Cout, _p(w), _p(m), _p(v), _p(dw), n, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group,
        Cout * KH * KW * Cin, 1 if hyper is not None else step, _p(hyper), lr, beta1, beta2, eps, _p(x_next), mode, _p(raw), _p(act),
        _p(gamma), _p(beta), gbs, _p(mean), _p(rstd), _p(sc_raw), _p(gamma_s), _p(beta_s), _p(mean_s), _p(rstd_s), _p(pooled),
        BN_EPS, _stream())
    if rc == _lib.MFT_EINVAL:
        return False
    _lib.check(rc, "mft_wgrad_adam_next_forward")
    return True


def conv2d_wgrad_adam_dgrad(x, dy, w, m, v, dxp, step, imgs_per_group, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, hyper=None):
    """Weight gradient + Adam of a per-episode 3x3/s1/p1 convolution that also leaves the data gradient as 9 per-tap partials
    in ``dxp`` [groups, 9, rows, Cin] (csrc/wgrad_dgrad.hip).  Returns False when the shape is outside the kernel's domain."""
    n, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    gs = w.shape[1] * w.shape[2]
    if hyper is not None:
        rc = _lib.lib().mft_conv2d_wgrad_adam_dgrad_nhwc_dev(_p(x), Cin, _p(dy), Cout, _p(w), _p(m), _p(v), _p(dxp), n, H, W, Cin,
                                                             Cout, imgs_per_group, gs, _p(hyper), beta1, beta2, eps, _stream())
    else:
        rc = _lib.lib().mft_conv2d_wgrad_adam_dgrad_nhwc(_p(x), Cin, _p(dy), Cout, _p(w), _p(m), _p(v), _p(dxp), n, H, W, Cin, Cout,
                                                         imgs_per_group, gs, step, lr, beta1, beta2, eps, _stream())
    if rc == _lib.MFT_EINVAL:
        return False
    _lib.check(rc, "mft_conv2d_wgrad_adam_dgrad_nhwc")
    return True


# ------------------------------------------------------------------------------------ batch norm

def bn_stats(x2d, C, rows_per_group, n_groups, running_mean=None, running_var=None, momentum=0.1, eps=BN_EPS,
             num_batches_tracked=None):
    """x2d [rows, ld] -> mean, rstd [n_groups, C].  ``num_batches_tracked`` (int64 scalar tensor, with running statistics only):
    incremented by the same launch."""
    _f32c(x2d)
    ld = x2d.shape[-1]
    mean = torch.empty((n_groups, C), device=x2d.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    nws = _lib.lib().mft_bn_stats_ws_floats(C, rows_per_group, n_groups)
    ws = torch.empty((max(int(nws), 1),), device=x2d.device, dtype=torch.float32)
    rc = _lib.lib().mft_bn_stats(_p(x2d), ld, C, rows_per_group, n_groups, eps, _p(mean), _p(rstd), _p(ws),
                                 _p(running_mean), _p(running_var), momentum, _p(num_batches_tracked), _stream())
    _lib.check(rc, "mft_bn_stats")
    return mean, rstd


BN_AFFINE_FMA = 0x100


def bn_apply(x2d, C, rows_per_group, n_groups, mean, rstd, gamma, beta, act=ACT_NONE, res=None, res_bn=None,
             out=None, gb_group_stride=0, slope=LRELU_SLOPE, planes=None, write_y=True, fma_affine=False):
    """y = act(bn(x) [+ res | + bn(res)]).  res_bn = (mean, rstd, gamma, beta) of the residual branch.
    ``planes``: int16 [3, rows, C] buffer that receives y split into its three bf16 pieces (operand of conv2d_x3p_bnstats);
    with ``write_y=False`` the fp32 y is not written at all.  ``fma_affine``: y = fma(x, rstd*gamma, beta - mean*rstd*gamma), the
    arithmetic of the frozen trunk's folded BatchNorm kernels (bit-identical to them)."""
    _f32c(x2d)
    if fma_affine:
        act = act | BN_AFFINE_FMA
    if out is None and (planes is None or write_y):
        out = torch.empty_like(x2d)
    rm = rr = rg = rb = None
    if res_bn is not None:
        rm, rr, rg, rb = res_bn
    if planes is not None:
        y = out if write_y else None
        rc = _lib.lib().mft_bn_apply_planes(_p(x2d), x2d.shape[-1], _p(y), 0 if y is None else y.shape[-1], _p(planes),
                                            planes.shape[1] * planes.shape[2], C, rows_per_group, n_groups, _p(mean), _p(rstd),
                                            _p(gamma), _p(beta), gb_group_stride, _p(res), 0 if res is None else res.shape[-1],
                                            _p(rm), _p(rr), _p(rg), _p(rb), act, slope, _stream())
        _lib.check(rc, "mft_bn_apply_planes")
        return y
    rc = _lib.lib().mft_bn_apply(_p(x2d), x2d.shape[-1], _p(out), out.shape[-1], C, rows_per_group, n_groups,
                                 _p(mean), _p(rstd), _p(gamma), _p(beta), gb_group_stride,
                                 _p(res), 0 if res is None else res.shape[-1], _p(rm), _p(rr), _p(rg), _p(rb),
                                 act, slope, _stream())
    _lib.check(rc, "mft_bn_apply")
    return out


class _BnStatsJob(ctypes.Structure):
    """MftBnStatsJob (include/mft_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "mean", "rstd", "ws", "running_mean", "running_var", "nbt")] + \
               [(n, ctypes.c_int) for n in ("ldx", "C", "rows_per_group", "n_groups")] + [("eps", ctypes.c_float), ("momentum", ctypes.c_float)]


def bn_stats_multi(jobs, momentum=0.1, eps=BN_EPS):
    """``jobs``: list of (x2d, C, rows_per_group, n_groups, running_mean | None, running_var | None, num_batches_tracked | None) --
    bn_stats of each in ONE launch pair (mft_bn_stats_multi, up to 8 jobs); returns [(mean, rstd), ...]."""
    arr = (_BnStatsJob * len(jobs))()
    out, keep = [], []
    for a, (x, C, rpg, ng, rm, rv, nbt) in zip(arr, jobs):
        _f32c(x)
        mean = torch.empty((ng, C), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ws = torch.empty((max(int(_lib.lib().mft_bn_stats_ws_floats(C, rpg, ng)), 1),), device=x.device, dtype=torch.float32)
        a.x, a.mean, a.rstd, a.ws = x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr()
        a.running_mean = None if rm is None else rm.data_ptr()
        a.running_var = None if rv is None else rv.data_ptr()
        a.nbt = None if nbt is None else nbt.data_ptr()
        a.ldx, a.C, a.rows_per_group, a.n_groups, a.eps, a.momentum = x.shape[-1], C, rpg, ng, eps, momentum
        out.append((mean, rstd))
        keep.append(ws)
    _lib.check(_lib.lib().mft_bn_stats_multi(arr, len(jobs), _stream(jobs[0][0].device)), "mft_bn_stats_multi")
    return out


class _BnFwdJob(ctypes.Structure):
    """MftBnFwdJob (include/mft_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "y", "gamma", "beta", "mean", "rstd", "running_mean", "running_var", "nbt", "res",
                                                "res_gamma", "res_beta", "res_mean", "res_rstd", "res_running_mean", "res_running_var",
                                                "res_nbt")] + \
               [(n, ctypes.c_int) for n in ("ldx", "ldy", "ldr", "C", "rows_per_group", "n_groups", "act")] + \
               [(n, ctypes.c_float) for n in ("eps", "momentum", "slope")]


def bn_forward_small_ok(C, rows_per_group):
    """The one-launch train-mode BatchNorm (mft_bn_forward_small) takes this problem: four-channel column groups, at most
    mft_bn_forward_small_max_rows() rows per group (512: the head's BatchNorm1d layers; a test hook can move it)."""
    return C % 4 == 0 and 0 < rows_per_group <= int(_lib.lib().mft_bn_forward_small_max_rows())


def bn_forward_small(x2d, C, rows_per_group, n_groups, gamma, beta, act=ACT_NONE, running=None, res=None, res_bn=None, out=None,
                     momentum=0.1, eps=BN_EPS, slope=LRELU_SLOPE, out_col=0):
    """bn_stats + bn_apply of a small tensor in ONE launch: -> (y, mean, rstd) -- or (y, mean, rstd, res_mean, res_rstd) when the
    residual goes through its own BatchNorm: ``res_bn`` = (gamma, beta, running | None).  ``running`` = (running_mean, running_var,
    num_batches_tracked) or None; group 0 advances them, as bn_stats does.  ``out_col``: write into columns out_col .. out_col + C of
    the (wider) matrix ``out``."""
    _f32c(x2d)
    dev = x2d.device
    if out is None:
        out = torch.empty_like(x2d)
    assert out_col >= 0 and out_col + C <= out.shape[-1]
    mean = torch.empty((n_groups, C), device=dev, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    j = _BnFwdJob()
    j.x, j.y, j.gamma, j.beta, j.mean, j.rstd = (t.data_ptr() for t in (x2d, out, gamma, beta, mean, rstd))
    j.y = out.data_ptr() + 4 * out_col
    rm, rv, nbt = running if running is not None else (None, None, None)
    j.running_mean, j.running_var, j.nbt = (None if t is None else t.data_ptr() for t in (rm, rv, nbt))
    rmean = rrstd = None
    if res is not None:
        _f32c(res)
        j.res, j.ldr = res.data_ptr(), res.shape[-1]
        if res_bn is not None:
            rg, rb, rrun = res_bn
            rmean = torch.empty((n_groups, C), device=dev, dtype=torch.float32)
            rrstd = torch.empty_like(rmean)
            j.res_gamma, j.res_beta, j.res_mean, j.res_rstd = rg.data_ptr(), rb.data_ptr(), rmean.data_ptr(), rrstd.data_ptr()
            rrm, rrv, rnbt = rrun if rrun is not None else (None, None, None)
            j.res_running_mean, j.res_running_var, j.res_nbt = (None if t is None else t.data_ptr() for t in (rrm, rrv, rnbt))
    j.ldx, j.ldy, j.C, j.rows_per_group, j.n_groups, j.act = x2d.shape[-1], out.shape[-1], C, rows_per_group, n_groups, act
    j.eps, j.momentum, j.slope = eps, momentum, slope
    _lib.check(_lib.lib().mft_bn_forward_small(ctypes.byref(j), _stream(dev)), "mft_bn_forward_small")
    if rmean is not None:
        return out, mean, rstd, rmean, rrstd
    return out, mean, rstd


class _BnBwdJob(ctypes.Structure):
    """MftBnBwdJob (include/mft_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "dy", "y_act", "dx", "mean", "rstd", "gamma", "dgamma", "dbeta", "ws", "dgamma_sum",
                                                "dbeta_sum", "dbias_zero")] + \
               [(n, ctypes.c_int) for n in ("ldx", "lddy", "ldya", "lddx", "C", "rows_per_group", "n_groups", "act")] + \
               [("slope", ctypes.c_float), ("reserved", ctypes.c_int)]


def bn_backward_multi(jobs):
    """``jobs``: list of (x2d, dy2d, y_act | None, C, rows, groups, mean, rstd, gamma, act) -- BatchNorm backward (+ activation
    derivative) of each in ONE launch triple (mft_bn_backward_act_multi); returns [(dx, dgamma [C], dbeta [C]), ...] with the
    parameter gradients summed over the groups."""
    arr = (_BnBwdJob * len(jobs))()
    out, keep = [], []
    for a, (x, dy, ya, C, rows, groups, mean, rstd, gamma, act) in zip(arr, jobs):
        _f32c(x)
        _f32c(dy)
        dev = x.device
        rpg = rows // groups
        dx = torch.empty_like(x)
        dg, db = torch.empty((C,), device=dev, dtype=torch.float32), torch.empty((C,), device=dev, dtype=torch.float32)
        ws = torch.empty((int(_lib.lib().mft_bn_backward_ws_floats(C, rpg, groups)),), device=dev, dtype=torch.float32)
        a.x, a.dy, a.dx, a.mean, a.rstd, a.gamma, a.ws = (t.data_ptr() for t in (x, dy, dx, mean, rstd, gamma, ws))
        a.y_act = None if ya is None else ya.data_ptr()
        one = groups == 1
        a.dgamma, a.dbeta = (dg.data_ptr(), db.data_ptr()) if one else (None, None)
        a.dgamma_sum, a.dbeta_sum = (None, None) if one else (dg.data_ptr(), db.data_ptr())
        a.dbias_zero = None
        a.ldx, a.lddy, a.ldya, a.lddx = x.shape[-1], dy.shape[-1], 0 if ya is None else ya.shape[-1], dx.shape[-1]
        a.C, a.rows_per_group, a.n_groups, a.act, a.slope = C, rpg, groups, act, LRELU_SLOPE
        out.append((dx, dg, db))
        keep.append(ws)
    _lib.check(_lib.lib().mft_bn_backward_act_multi(arr, len(jobs), _stream(jobs[0][0].device)), "mft_bn_backward_act_multi")
    return out


class _BnApplyJob(ctypes.Structure):
    """MftBnApplyJob (include/mft_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "y", "mean", "rstd", "gamma", "beta")] + \
               [(n, ctypes.c_int) for n in ("ldx", "ldy", "C", "rows_per_group", "n_groups", "act")] + [("slope", ctypes.c_float), ("reserved", ctypes.c_int)]


def bn_apply_multi(jobs):
    """``jobs``: list of (x2d, C, rows_per_group, n_groups, mean, rstd, gamma, beta, act, out) -- bn_apply of each (no residual) in ONE
    launch per 16 jobs (mft_bn_apply_multi); returns the list of outputs."""
    if not jobs:
        return []
    arr = (_BnApplyJob * len(jobs))()
    for a, (x, C, rpg, ng, mean, rstd, gamma, beta, act, out) in zip(arr, jobs):
        _f32c(x)
        a.x, a.y, a.mean, a.rstd, a.gamma, a.beta = (t.data_ptr() for t in (x, out, mean, rstd, gamma, beta))
        a.ldx, a.ldy, a.C, a.rows_per_group, a.n_groups, a.act, a.slope = x.shape[-1], out.shape[-1], C, rpg, ng, act, LRELU_SLOPE
    _lib.check(_lib.lib().mft_bn_apply_multi(arr, len(jobs), _stream(jobs[0][0].device)), "mft_bn_apply_multi")
    return [j[-1] for j in jobs]


def bn_relu_maxpool(x, mean, rstd, gamma, beta, imgs_per_group=0):
    _f32c(x)
    n, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((n, OH, OW, C), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mft_bn_relu_maxpool(_p(x), _p(y), n, H, W, C, imgs_per_group, _p(mean), _p(rstd), _p(gamma),
                                        _p(beta), _stream())
    _lib.check(rc, "mft_bn_relu_maxpool")
    return y


def bn_relu_maxpool_gather(cache, src_idx, n_img, mean, rstd, gamma, beta, imgs_per_group, out=None, planes=None):
    """bn_relu_maxpool over images cache[src_idx[n]] (cache [slots,H,W,C]); n_img = src_idx.numel().
    ``planes``: optional int16 [3, n_img*OH*OW, C] buffer receiving the bf16x3 split of the output as well."""
    _f32c(cache)
    _, H, W, C = cache.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    if out is None:
        out = torch.empty((n_img, OH, OW, C), device=cache.device, dtype=torch.float32)
    if planes is not None:
        rc = _lib.lib().mft_bn_relu_maxpool_gather_planes(_p(cache), _p(src_idx), _p(out), _p(planes),
                                                          planes.shape[1] * planes.shape[2], n_img, H, W, C, imgs_per_group,
                                                          _p(mean), _p(rstd), _p(gamma), _p(beta), _stream())
        _lib.check(rc, "mft_bn_relu_maxpool_gather_planes")
        return out
    rc = _lib.lib().mft_bn_relu_maxpool_gather(_p(cache), _p(src_idx), _p(out), n_img, H, W, C, imgs_per_group, _p(mean),
                                               _p(rstd), _p(gamma), _p(beta), _stream())
    _lib.check(rc, "mft_bn_relu_maxpool_gather")
    return out


def bn_image_moments(x):
    """x [n_img,H,W,C] -> per-image (mean [n_img,C], M2 [n_img,C]) over the H*W pixels."""
    _f32c(x)
    n, H, W, C = x.shape
    mean = torch.empty((n, C), device=x.device, dtype=torch.float32)
    m2 = torch.empty_like(mean)
    _lib.check(_lib.lib().mft_bn_image_moments(_p(x), C, C, H * W, n, _p(mean), _p(m2), _stream()), "mft_bn_image_moments")
    return mean, m2


def bn_combine_moments(mean_img, m2_img, idx_i32, rows_per_img, imgs_per_group, n_groups, mean=None, rstd=None, eps=BN_EPS):
    """Mini-batch BatchNorm statistics of the groups of images idx[g*k:(g+1)*k] from cached per-image moments."""
    C = mean_img.shape[1]
    if mean is None:
        mean = torch.empty((n_groups, C), device=mean_img.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
    rc = _lib.lib().mft_bn_combine_moments(_p(mean_img), _p(m2_img), _p(idx_i32), C, rows_per_img, imgs_per_group,
                                           n_groups, eps, _p(mean), _p(rstd), _stream())
    _lib.check(rc, "mft_bn_combine_moments")
    return mean, rstd


def global_avgpool(x):
    _f32c(x)
    n, H, W, C = x.shape
    y = torch.empty((n, C), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_global_avgpool(_p(x), _p(y), n, H * W, C, _stream()), "mft_global_avgpool")
    return y


def avgpool_relu_backward(dfeat, out):
    _f32c(dfeat)
    _f32c(out)
    n, H, W, C = out.shape
    d = torch.empty_like(out)
    _lib.check(_lib.lib().mft_avgpool_relu_backward(_p(dfeat), _p(out), _p(d), n, H * W, C, _stream()),
               "mft_avgpool_relu_backward")
    return d


def bn_backward(x2d, dy2d, C, rows_per_group, n_groups, mean, rstd, gamma, relu_out=None, need_dx=True,
                gb_group_stride=0):
    """-> dx [rows,C] (or None), dgamma [n_groups,C], dbeta [n_groups,C]."""
    _f32c(x2d)
    _f32c(dy2d)
    dx = torch.empty_like(x2d) if need_dx else None
    dg = torch.empty((n_groups, C), device=x2d.device, dtype=torch.float32)
    db = torch.empty_like(dg)
    rc = _lib.lib().mft_bn_backward(_p(x2d), x2d.shape[-1], _p(dy2d), dy2d.shape[-1], _p(relu_out),
                                    0 if relu_out is None else relu_out.shape[-1], _p(dx),
                                    0 if dx is None else dx.shape[-1], C, rows_per_group, n_groups, _p(mean), _p(rstd),
                                    _p(gamma), gb_group_stride, _p(dg), _p(db), _stream())
    _lib.check(rc, "mft_bn_backward")
    return dx, dg, db


# ------------------------------------------------------------------------------------ loss / optim

def cross_entropy(logits, labels_i32, rows_per_group, n_groups, need_grad=True):
    """-> loss [n_groups], dlogits (or None)."""
    _f32c(logits)
    rows, C = logits.shape
    buf = torch.empty((n_groups + rows,), device=logits.device, dtype=torch.float32)
    d = torch.empty_like(logits) if need_grad else None
    rc = _lib.lib().mft_cross_entropy(_p(logits), C, _p(labels_i32), C, rows_per_group, n_groups, _p(buf), _p(d),
                                      _stream())
    _lib.check(rc, "mft_cross_entropy")
    return buf[:n_groups], d


def softmax_rows(x):
    _f32c(x)
    rows, C = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().mft_softmax_rows(_p(x), C, _p(y), C, C, rows, _stream()), "mft_softmax_rows")
    return y


def adam_hyper_advance(step_i32, hyper, lr=0.01, beta1=0.9, beta2=0.999):
    """Device-side t = ++step; hyper = {lr/(1-beta1^t), 1/sqrt(1-beta2^t)} (one tiny launch; graph-capturable)."""
    _lib.check(_lib.lib().mft_adam_hyper_advance(_p(step_i32), _p(hyper), lr, beta1, beta2, _stream()), "mft_adam_hyper_advance")


def adam_step(p, g, m, v, step, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, hyper=None):
    """In-place fused Adam on flat (contiguous) fp32 tensors of equal numel."""
    for t in (p, g, m, v):
        _f32c(t)
    if hyper is not None:
        rc = _lib.lib().mft_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), beta1, beta2, eps, weight_decay,
                                          _stream())
        _lib.check(rc, "mft_adam_step_dev")
        return
    rc = _lib.lib().mft_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), step, lr, beta1, beta2, eps, weight_decay,
                                  _stream())
    _lib.check(rc, "mft_adam_step")


def sgd_step(p, g, buf, first_step, lr=0.01, momentum=0.9, dampening=0.9, weight_decay=0.001):
    rc = _lib.lib().mft_sgd_step(_p(p), _p(g), _p(buf), p.numel(), 1 if first_step else 0, lr, momentum, dampening,
                                 weight_decay, _stream())
    _lib.check(rc, "mft_sgd_step")


def maml_delta(p, p2, p3):
    _lib.check(_lib.lib().mft_maml_delta(_p(p), _p(p2), _p(p3), p.numel(), _stream()), "mft_maml_delta")


# ------------------------------------------------------------------------------------ gnn glue

def pair_absdiff(x, N, F, ldd):
    """x [n_graphs*N, ldx] -> d [n_graphs*N*N, ldd] = |x_i - x_j| (zero padded beyond F)."""
    _f32c(x)
    n_graphs = x.shape[0] // N
    d = torch.empty((n_graphs * N * N, ldd), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_pair_absdiff(_p(x), x.shape[1], _p(d), ldd, n_graphs, N, F, _stream()),
               "mft_pair_absdiff")
    return d


def masked_softmax(s, N):
    """s [n_graphs*N*N, lds] (column 0) -> A [n_graphs, N, N]."""
    _f32c(s)
    n_graphs = s.shape[0] // (N * N)
    A = torch.empty((n_graphs, N, N), device=s.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_masked_softmax(_p(s), s.shape[1], _p(A), n_graphs, N, _stream()), "mft_masked_softmax")
    return A


def graph_aggregate(A, x, F, ldy):
    """-> y [n_graphs*N, ldy] = cat(x[:, :F], A@x[:, :F]) zero padded."""
    _f32c(A)
    _f32c(x)
    n_graphs, N, _ = A.shape
    y = torch.empty((n_graphs * N, ldy), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_graph_aggregate(_p(A), _p(x), x.shape[1], _p(y), ldy, n_graphs, N, F, _stream()),
               "mft_graph_aggregate")
    return y


def copy_cols(x, y, col_off, C, act=ACT_NONE, slope=LRELU_SLOPE):
    _lib.check(_lib.lib().mft_copy_cols(_p(x), x.shape[1], _p(y), y.shape[1], col_off, C, x.shape[0], act, slope,
                                        _stream()), "mft_copy_cols")
    return y


def build_graph_nodes(z, n_episodes, n_way, n_support, n_query, ld=256, fold=False):
    """z [n_episodes*n_way*(S+n_query), 128] -> nodes [n_episodes*n_query*n_way*(n_support+1), ld]."""
    _f32c(z)
    rows = n_episodes * n_query * n_way * (n_support + 1)
    nodes = torch.empty((rows, ld), device=z.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_build_graph_nodes(_p(z), z.shape[1], _p(nodes), ld, n_episodes, n_way, n_support,
                                                n_query, 1 if fold else 0, _stream()), "mft_build_graph_nodes")
    return nodes


def gather_query_scores(out, n_episodes, n_way, n_support, n_query):
    _f32c(out)
    scores = torch.empty((n_episodes * n_way * n_query, n_way), device=out.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_gather_query_scores(_p(out), out.shape[1], _p(scores), n_episodes, n_way, n_support,
                                                  n_query, _stream()), "mft_gather_query_scores")
    return scores


def gather_rows(src2d, idx_i32, out=None):
    """dst[r] = src2d[idx[r]] for rows of src2d.shape[1] floats."""
    _f32c(src2d)
    n = idx_i32.numel()
    if out is None:
        out = torch.empty((n, src2d.shape[1]), device=src2d.device, dtype=torch.float32)
    _lib.check(_lib.lib().mft_gather_rows(_p(src2d), _p(idx_i32), _p(out), n, src2d.shape[1], _stream()),
               "mft_gather_rows")
    return out
