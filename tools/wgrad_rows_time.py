#!/usr/bin/env python3
"""Standalone time of the three fused weight-gradient + Adam launches of the last block at E episodes:
32 x 128 stream-shaped kernel (wgrad_adam_rows_kernel, default) vs the 64 x 64 tile kernel (mft_debug_set_conv_tile(9500)),
and a bit-identity check of the two.  Usage: wgrad_rows_time.py [E]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
lib = _lib.lib()
g = torch.Generator(device=dev); g.manual_seed(1)
shapes = [("trunk.7.C2", 512, 512, 3, 1, 1, 3), ("trunk.7.C1", 256, 512, 3, 2, 1, 6), ("trunk.7.shortcut", 256, 512, 1, 2, 0, 6)]
for name, Cin, Cout, k, stride, pad, H in shapes:
    OH = (H + 2 * pad - k) // stride + 1
    x = torch.randn(E * 5, H, H, Cin, device=dev, generator=g)
    dy = torch.randn(E * 5, OH, OH, Cout, device=dev, generator=g) * 1e-3
    K = k * k * Cin
    w0 = torch.randn(E, Cout, K, device=dev, generator=g) * 0.02
    res = {}
    for knob, tag in ((9501, "rows 32x128"), (9504, "rows walking"), (9505, "rows co-walk"), (9500, "tile 64x64")):
        lib.mft_debug_set_conv_tile(knob)
        w, m, v = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
        ops.conv2d_wgrad_adam(x, dy, w, m, v, Cout, k, k, stride, pad, 1, 5)
        torch.cuda.synchronize()
        res[tag] = (w.clone(), m.clone(), v.clone())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for it in range(10):
            ops.conv2d_wgrad_adam(x, dy, w, m, v, Cout, k, k, stride, pad, 2 + it, 5)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        print("%-18s %-12s %8.1f us  %.2f TB/s" % (name, tag, us, 24.0 * w.numel() / us / 1e6))
    same = all(torch.equal(p, q) for p, q in zip(res["rows 32x128"], res["rows walking"]))
    same2 = all(torch.equal(p, q) for p, q in zip(res["rows 32x128"], res["rows co-walk"]))
    print("%-18s bit-identical w/m/v after step 1: walking %s, co-walk %s" % (name, same, same2))
lib.mft_debug_reset()
