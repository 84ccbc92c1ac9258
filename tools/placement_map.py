#!/usr/bin/env python3
"""Map of the 3R+3W stream rate over the device's address space: K consecutive slab-sized allocations (1.88 GB each), the rate of
every triple of NEIGHBOURS (i, i+1, i+2) and of a few spread triples.  Question (round 5): on a lease whose best-of-12 placement
is 6.0 TB/s instead of 6.4, is there a better region further out?   Usage: placement_map.py [K] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 128 * 3673088 // 1024 * 1024
lib = _lib.lib()
t0 = time.perf_counter()
bufs = []
for i in range(K):
    bufs.append(torch.empty(n, device="cuda:0"))
torch.cuda.synchronize()
t1 = time.perf_counter()
for b in bufs:
    b.zero_()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("allocated %d x %.2f GB in %.2f s, zeroed in %.2f s" % (K, n * 4 / 1e9, t1 - t0, t2 - t1))
print("address >> 30 of the buffers:", [b.data_ptr() >> 30 for b in bufs])


def rate(t):
    w, m, v = (bufs[i] for i in t)
    lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
    b.record()
    torch.cuda.synchronize()
    return 24.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e12


rate((0, 1, 2))
nb = [rate((i, i + 1, i + 2)) for i in range(K - 2)]
print("neighbour triples (i, i+1, i+2), TB/s:")
for i in range(0, K - 2, 10):
    print("  %3d: " % i + " ".join("%.2f" % r for r in nb[i:i + 10]))
nb2 = [rate((i, i + 1, i + 2)) for i in range(K - 2)]
print("second sweep, |difference|: median %.3f max %.3f" % (np.median(np.abs(np.array(nb) - nb2)), np.max(np.abs(np.array(nb) - nb2))))
best = int(np.argmax(nb))
print("best neighbour triple at %d: %.2f; worst %.2f; median %.2f" % (best, max(nb), min(nb), np.median(nb)))
rs = np.random.RandomState(0)
sp = [tuple(sorted(rs.choice(K, 3, replace=False))) for _ in range(60)]
spr = [rate(t) for t in sp]
o = np.argsort(spr)[::-1]
print("60 random triples: max %.2f median %.2f min %.2f; best: %s" % (max(spr), np.median(spr), min(spr), [(sp[i], "%.2f" % spr[i]) for i in o[:5]]))
# around the best neighbour triple: all triples of the 8 buffers starting there
import itertools
lo = max(0, min(best - 2, K - 8))
loc = {t: rate(t) for t in itertools.combinations(range(lo, lo + 8), 3)}
v = np.array(list(loc.values()))
print("all 56 triples of buffers %d..%d: max %.2f median %.2f min %.2f" % (lo, lo + 7, v.max(), np.median(v), v.min()))
