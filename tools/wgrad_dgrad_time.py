#!/usr/bin/env python3
"""trunk.7.C2 backward at E episodes: [data gradient + BN1 backward] + [weight gradient + Adam] as two passes over the weights
vs the one-pass kernel (csrc/wgrad_dgrad.hip) + col2im/BN1 backward.  Usage: wgrad_dgrad_time.py [E]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops, _lib

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, C = E * 5, 512
lib = _lib.lib()
r1 = torch.relu(torch.randn(n, 3, 3, C, device="cuda"))
c1 = torch.randn(n, 3, 3, C, device="cuda")
dc2 = torch.randn(n, 3, 3, C, device="cuda") * 1e-3
w = torch.randn(E, C, 9 * C, device="cuda") * 0.02
m, v = torch.zeros_like(w), torch.zeros_like(w)
g1 = torch.ones(E, C, device="cuda")
mean, rstd = ops.bn_stats(c1.view(-1, C), C, 45, E)
dc1 = torch.empty_like(c1)
dg, db = torch.empty(E, C, device="cuda"), torch.empty(E, C, device="cuda")
dxp = torch.empty(E, 9, 45, C, device="cuda")


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def dgrad_bn():
    assert lib.mft_conv2d_dgrad_bn_backward_small(ops._p(dc2), C, ops._p(w), ops._p(dc1), C, n, 3, 3, C, C, 3, 3, 1, 5, C * 9 * C,
                                                  ops._p(c1), ops._p(r1), ops._p(mean), ops._p(rstd), ops._p(g1), C, ops._p(dg),
                                                  ops._p(db), ops._stream()) == 0


def wgrad():
    ops.conv2d_wgrad_adam(r1, dc2, w, m, v, C, 3, 3, 1, 1, 3, imgs_per_group=5)


def fused():
    assert ops.conv2d_wgrad_adam_dgrad(r1, dc2, w, m, v, dxp, 3, 5)


def col2im():
    assert lib.mft_col2im_bn_backward_small(ops._p(dxp), ops._p(c1), ops._p(r1), ops._p(dc1), n, 3, 3, C, 5, ops._p(mean),
                                            ops._p(rstd), ops._p(g1), C, ops._p(dg), ops._p(db), ops._stream()) == 0


a, b, c, d = t(dgrad_bn), t(wgrad), t(fused), t(col2im)
byt = 24.0 * w.numel()
print("E=%d  two passes: dgrad+BN %.0f us + wgrad/Adam %.0f us (%.2f TB/s) = %.0f us | one pass: %.0f us (%.2f TB/s) + col2im/BN %.0f us = %.0f us"
      % (E, a, b, byt / b / 1e6, a + b, c, byt / c / 1e6, d, c + d))
