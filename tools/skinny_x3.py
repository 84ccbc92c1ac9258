#!/usr/bin/env python3
"""fp32-MFMA vs bf16x3 forms of the weight-streaming last-block kernels (csrc/skinny.hip), E episodes x 5 images.
Usage: python tools/skinny_x3.py [E]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = E * 5
lib = _lib.lib()


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


x6 = torch.randn(n, 6, 6, 256, device="cuda")
x3 = torch.randn(n, 3, 3, 512, device="cuda")
w1 = torch.randn(E, 512, 2304, device="cuda") * 0.02
w2 = torch.randn(E, 512, 4608, device="cuda") * 0.02
ws = torch.randn(E, 512, 256, device="cuda") * 0.05
o = torch.empty(n, 3, 3, 512, device="cuda")
cases = [
    ("C1 fwd (3x3 s2, 256->512)", lambda: ops.conv2d(x6, w1, 512, 3, 3, 2, 1, imgs_per_group=5, out=o), w1),
    ("C2 fwd (3x3 s1, 512->512)", lambda: ops.conv2d(x3, w2, 512, 3, 3, 1, 1, imgs_per_group=5, out=o), w2),
    ("shortcut fwd (1x1 s2)", lambda: ops.conv2d(x6, ws, 512, 1, 1, 2, 0, imgs_per_group=5, out=o), ws),
    ("C2 dgrad", lambda: ops.conv2d_dgrad(x3, w2, 512, 3, 3, 1, imgs_per_group=5, out=o), w2),
]
for name, fn, w in cases:
    res = []
    outs = []
    for mode in (8000, 8001):
        lib.mft_debug_set_conv_tile(mode)
        t = timeit(fn)
        outs.append(o.clone())
        res.append("%s %6.1f us (%.2f TB/s)" % ("x3  " if mode == 8001 else "fp32", t, 4.0 * w.numel() / t / 1e6))
    d = float((outs[0] - outs[1]).abs().max()) / float(outs[0].abs().max())
    print("%-28s %s | %s | max rel diff %.2e" % (name, res[0], res[1], d))
