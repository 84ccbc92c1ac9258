#!/usr/bin/env python3
"""Two layouts of the 12 slab-placement candidates compared on the same lease: A = three groups of four neighbouring allocations,
11 GB of ballast between the groups (the engine's scheme); B = twelve allocations 9.4 GB apart from each other.  60 consecutive
1.88 GB allocations stand in for both (A = indices 0-3, 10-13, 20-23; B = 0, 5, ..., 55); all 220 triples of each are timed with
the Adam-shaped probe, and pick_slab_buffers' choice is reported.   Usage: placement_layouts.py"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine, ops

n = 128 * 3673088 // 1024 * 1024
lib = _lib.lib()
bufs = [torch.empty(n, device="cuda:0") for _ in range(60)]
for b in bufs:
    b.zero_()


def rate(t, reps=2):
    w, m, v = (bufs[i] for i in t)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
    b.record()
    torch.cuda.synchronize()
    return 24.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e9


rate((0, 1, 2), 4)
for name, idx in (("A groups of four", [0, 1, 2, 3, 10, 11, 12, 13, 20, 21, 22, 23]), ("B all apart", list(range(0, 60, 5))),
                  ("A groups of four (again)", [0, 1, 2, 3, 10, 11, 12, 13, 20, 21, 22, 23])):
    rates = {t: rate(tuple(idx[i] for i in t)) for t in itertools.combinations(range(12), 3)}
    v = np.array(list(rates.values()))
    w_, m_, v_, w2_ = engine.pick_slab_buffers(rates, 12)
    print("%-26s best %.0f  p90 %.0f  median %.0f  worst %.0f GB/s; chosen (w, m, v) %.0f, (w2, m, v) %.0f; first three %.0f"
          % (name, v.max(), np.percentile(v, 90), np.median(v), v.min(), rates[tuple(sorted((w_, m_, v_)))],
             rates[tuple(sorted((w2_, m_, v_)))], rates[(0, 1, 2)]))
