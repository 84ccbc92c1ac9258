#!/usr/bin/env python3
"""Time every forward tile of mft_conv2d_nhwc on the ResNet10 layer shapes of one lockstep inner step
(E episodes x 5 images, 84x84).  Usage: python tools/conv_tune.py [E]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops, _lib

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = E * 5
LAYERS = [("trunk.4.C1", 64, 64, 3, 1, 1, 21), ("trunk.5.C1", 64, 128, 3, 2, 1, 21), ("trunk.5.C2", 128, 128, 3, 1, 1, 11),
          ("trunk.5.sc", 64, 128, 1, 2, 0, 21), ("trunk.6.C1", 128, 256, 3, 2, 1, 11), ("trunk.6.C2", 256, 256, 3, 1, 1, 6),
          ("trunk.6.sc", 128, 256, 1, 2, 0, 11)]
names = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x128", 4: "64x64"}
for (name, cin, cout, k, s, p, H) in LAYERS:
    x = torch.randn(n, H, H, cin, device="cuda")
    w = torch.randn(cout, k * k * cin, device="cuda") * 0.05
    OH = (H + 2 * p - k) // s + 1
    fl = 2.0 * n * OH * OH * cout * k * k * cin
    res = []
    for tile in (0, 1, 2, 3, 4):
        if tile in (1, 3) and cout % 128:
            continue
        _lib.lib().mft_debug_set_conv_tile(tile)
        out = ops.conv2d(x, w, cout, k, k, s, p)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.conv2d(x, w, cout, k, k, s, p, out=out)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        res.append("%s %.0fus %.0fTF" % (names[tile], us, fl / us / 1e6))
    _lib.lib().mft_debug_set_conv_tile(0)
    print("%-12s M=%-7d N=%-4d K=%-5d | %s" % (name, n * OH * OH, cout, k * k * cin, " | ".join(res)))

print("per-episode (grouped) weights, 5 images per group: generic tiles vs skinny kernel")
GL = [("trunk.7.C1", 256, 512, 3, 2, 1, 6), ("trunk.7.C2", 512, 512, 3, 1, 1, 3), ("trunk.7.sc", 256, 512, 1, 2, 0, 6)]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for (name, cin, cout, k, s, p, H) in GL:
    x = torch.randn(n, H, H, cin, device="cuda")
    w = torch.randn(E, cout, k * k * cin, device="cuda") * 0.05
    OH = (H + 2 * p - k) // s + 1
    byt = 4.0 * E * cout * k * k * cin
    res = []
    for mode, nm in ((3000, "generic"), (3001, "skinny")):
        _lib.lib().mft_debug_set_conv_tile(mode)
        out = ops.conv2d(x, w, cout, k, k, s, p, imgs_per_group=5)
        us = timeit(lambda: ops.conv2d(x, w, cout, k, k, s, p, imgs_per_group=5, out=out))
        res.append("fwd %s %.0fus %.2fTB/s" % (nm, us, byt / us / 1e6))
        if s == 1:
            dy = torch.randn(n, OH, OH, cout, device="cuda")
            dx = ops.conv2d_dgrad(dy, w, cin, k, k, p, imgs_per_group=5)
            us = timeit(lambda: ops.conv2d_dgrad(dy, w, cin, k, k, p, imgs_per_group=5, out=dx))
            res.append("dgrad %s %.0fus %.2fTB/s" % (nm, us, byt / us / 1e6))
    _lib.lib().mft_debug_set_conv_tile(3001)
    print("%-12s rows/group=%-4d N=%-4d K=%-5d | %s" % (name, 5 * OH * OH, cout, k * k * cin, " | ".join(res)))
