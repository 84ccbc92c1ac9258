#!/usr/bin/env python3
"""bf16x3 convolution vs the fp32-MFMA convolution on the trunk shapes: time, TFLOP/s (fp32-equivalent) and error against float64.
Usage: python tools/x3_tune.py [E]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops, _lib

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = E * 5
LAYERS = [("trunk.4.C1", 64, 64, 3, 1, 1, 21), ("trunk.5.C1", 64, 128, 3, 2, 1, 21), ("trunk.5.C2", 128, 128, 3, 1, 1, 11),
          ("trunk.5.sc", 64, 128, 1, 2, 0, 21), ("trunk.6.C1", 128, 256, 3, 2, 1, 11), ("trunk.6.C2", 256, 256, 3, 1, 1, 6),
          ("trunk.6.sc", 128, 256, 1, 2, 0, 11)]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


FILTER = sys.argv[2] if len(sys.argv) > 2 else ""
for (name, cin, cout, k, s, p, H) in LAYERS:
    if FILTER and FILTER not in name:
        continue
    x = torch.randn(n, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") * (2.0 / (k * k * cout)) ** 0.5
    wpk = ops.pack_conv_weight(w)
    w3 = ops.split_weight_x3(wpk)
    OH = (H + 2 * p - k) // s + 1
    fl = 2.0 * n * OH * OH * cout * k * k * cin
    out32 = ops.conv2d(x, wpk, cout, k, k, s, p)
    t32 = timeit(lambda: ops.conv2d(x, wpk, cout, k, k, s, p, out=out32))
    res = ["fp32-mfma %.0fus %.0fTF" % (t32, fl / t32 / 1e6)]
    nv = 8
    ref = F.conv2d(x[:nv].permute(0, 3, 1, 2).double(), w.double(), None, s, p).permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    e32 = float((out32[:nv].double() - ref).abs().max()) / sc
    for tile, tn in ((1, "128x64"), (2, "128x128"), (11, "patch")):
        if tile == 2 and cout % 128:
            continue
        if tile == 11 and not (k == 3 and s == 1):
            continue
        _lib.lib().mft_debug_set_x3_tile(10 if tile < 10 else 12)
        _lib.lib().mft_debug_set_x3_tile(tile if tile < 10 else 0)
        o3 = ops.conv2d_x3(x, w3, cout, k, k, s, p)
        t3 = timeit(lambda: ops.conv2d_x3(x, w3, cout, k, k, s, p, out=o3))
        e3 = float((o3[:nv].double() - ref).abs().max()) / sc
        res.append("x3 %s %.0fus %.0fTF err %.1e" % (tn, t3, fl / t3 / 1e6, e3))
    _lib.lib().mft_debug_set_x3_tile(0)
    _lib.lib().mft_debug_set_x3_tile(11)
    print("%-11s M=%-7d N=%-4d K=%-5d | %s (err %.1e) | %s" % (name, n * OH * OH, cout, k * k * cin, res[0], e32, " | ".join(res[1:])))
