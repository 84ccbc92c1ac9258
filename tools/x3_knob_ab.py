#!/usr/bin/env python3
"""Standalone time of the trunk's bf16x3 convolutions under mft_debug_set_x3_tile knobs.  Usage: x3_knob_ab.py knob [knob ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops, _lib

E = 128
n = E * 5
LAYERS = [("trunk.4.C1", 64, 64, 3, 1, 1, 21), ("trunk.5.C1", 64, 128, 3, 2, 1, 21), ("trunk.5.C2", 128, 128, 3, 1, 1, 11),
          ("trunk.6.C1", 128, 256, 3, 2, 1, 11), ("trunk.6.C2", 256, 256, 3, 1, 1, 6)]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


knobs = [int(k) for k in sys.argv[1:]] or [41]
data = []
for (name, cin, cout, k, s, p, H) in LAYERS:
    x = torch.randn(n, H, H, cin, device="cuda")
    split = ops.split_weight_h2 if os.environ.get("X3_H2", "1") == "1" else ops.split_weight_x3      # f16x2 (default) | bf16x3 planes
    w3 = split(ops.pack_conv_weight(torch.randn(cout, cin, k, k, device="cuda") * 0.05))
    data.append((name, x, w3, cout, k, s, p, ops.conv2d_x3(x, w3, cout, k, k, s, p)))
ref = [d[7].clone() for d in data]            # outputs under the default knobs
for kn in knobs:
    _lib.lib().mft_debug_set_x3_tile(kn)
    ts = [timeit(lambda: ops.conv2d_x3(x, w3, cout, k, k, s, p, out=o)) for (name, x, w3, cout, k, s, p, o) in data]
    same = all(torch.equal(d[7], r) for d, r in zip(data, ref))
    print("         bit-identical to the default form: %s" % same)
    print("knob %3d: " % kn + "  ".join("%s %.0f" % (d[0], t) for d, t in zip(data, ts)) + "  | sum %.0f us" % sum(ts))
