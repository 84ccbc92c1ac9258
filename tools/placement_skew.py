#!/usr/bin/env python3
"""Is the 3R+3W stream's rate a function of the RELATIVE placement of its three arrays?  One big allocation; w at offset 0, m at
S + d1, v at 2 S + d2 (S = slab size rounded up to 2 MB) for a grid of skews d1, d2.  Round 5: random far-apart triples of separate
allocations stream at 6.30 TB/s (median) where neighbouring allocations get 5.36 (tools/placement_map.py).
Usage: placement_skew.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 128 * 3673088 // 1024 * 1024
lib = _lib.lib()
S = (n * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20)
PAD = 1 << 30
big = torch.empty((3 * S + 2 * PAD) // 4 + 1024, device="cuda:0")
big.zero_()
base = big.data_ptr()
print("base %x, S = %d MB" % (base, S >> 20))
import ctypes


def rate(offs):
    p = [ctypes.c_void_p(base + o) for o in offs]
    lib.mft_stream_probe(p[0], p[1], p[2], n, ops._stream(big))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.mft_stream_probe(p[0], p[1], p[2], n, ops._stream(big))
    b.record()
    torch.cuda.synchronize()
    return 24.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e12


rate((0, S, 2 * S))
skews = [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 3 << 19, 1 << 22, 1 << 24, 5 << 22, 1 << 26, 3 << 25, 1 << 28, 1 << 29]
print("rate(w at 0, m at S + d1, v at 2S + d2), TB/s; rows d1, columns d2 = same list")
print("d:", skews)
tab = np.zeros((len(skews), len(skews)))
for i, d1 in enumerate(skews):
    for j, d2 in enumerate(skews):
        tab[i, j] = rate((0, S + d1, 2 * S + d2))
    print("%10d: " % d1 + " ".join("%.2f" % v for v in tab[i]))
print("max %.2f at %s; min %.2f; median %.2f" % (tab.max(), np.unravel_index(tab.argmax(), tab.shape), tab.min(), np.median(tab)))
