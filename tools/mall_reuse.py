#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache help when the C2 dgrad (reads w) and the fused C2 wgrad+Adam (reads w again) are run
back-to-back on chunks of episodes whose weights fit on-die, instead of layer-by-layer over all E episodes?
Usage: python tools/mall_reuse.py [E]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = E * 5
r1 = torch.randn(n, 3, 3, 512, device="cuda")
dc2 = torch.randn(n, 3, 3, 512, device="cuda") * 1e-3
dr1 = torch.empty(n, 3, 3, 512, device="cuda")
o7 = torch.empty(n, 3, 3, 512, device="cuda")
w = torch.randn(E, 512, 4608, device="cuda") * 0.02
m = torch.zeros_like(w)
v = torch.zeros_like(w)


def pair(lo, hi, fwd):
    a, b = lo * 5, hi * 5
    if fwd:
        ops.conv2d(r1[a:b], w[lo:hi], 512, 3, 3, 1, 1, imgs_per_group=5, out=o7[a:b])
    ops.conv2d_dgrad(dc2[a:b], w[lo:hi], 512, 3, 3, 1, imgs_per_group=5, out=dr1[a:b])
    ops.conv2d_wgrad_adam(r1[a:b], dc2[a:b], w[lo:hi], m[lo:hi], v[lo:hi], 512, 3, 3, 1, 1, 3, imgs_per_group=5)


def run(chunk, fwd, iters=10):
    def once():
        for lo in range(0, E, chunk):
            pair(lo, min(E, lo + chunk), fwd)
    once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        once()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for fwd in (False, True):
    print("== %s ==" % ("fwd + dgrad + wgrad/Adam of trunk.7.C2" if fwd else "dgrad + wgrad/Adam of trunk.7.C2"))
    for chunk in (E, 64, 32, 16, 8, 4, 2):
        if chunk > E:
            continue
        t = run(chunk, fwd)
        byt = (8 if fwd else 7) * 4.0 * w.numel()
        print("chunk %3d episodes (%5.1f MB of w per chunk): %7.0f us  (%.2f TB/s algorithmic)" % (
            chunk, chunk * 512 * 4608 * 4 / 1e6, t, byt / t / 1e6))
