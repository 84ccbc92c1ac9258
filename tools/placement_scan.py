#!/usr/bin/env python3
"""Is the rate of the 3R+3W stream a property of WHERE its three arrays live?  Allocates K separate slab-sized buffers (128 episodes x
3,673,088 floats = 1.88 GB each), times mft_stream_probe over every triple of them and prints the spread, the per-buffer means and the
best / worst triples.   Usage: placement_scan.py [K] [n_floats]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128 * 3673088 // 1024 * 1024
dev = "cuda:0"
lib = _lib.lib()
bufs = [torch.zeros(n, device=dev) for _ in range(K)]
print("buffers:", ["%x" % (b.data_ptr() >> 21) for b in bufs], "(address >> 21)")


def rate(t, reps=4):
    w, m, v = (bufs[i] for i in t)
    lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
    b.record(); torch.cuda.synchronize()
    return 24.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e12


res = {}
for rnd in range(2):                                  # two sweeps: is a triple's rate reproducible?
    for t in itertools.combinations(range(K), 3):
        res.setdefault(t, []).append(rate(t))
r = np.array([np.mean(v) for v in res.values()])
d = np.array([abs(v[0] - v[1]) for v in res.values()])
print("%d triples: %.2f .. %.2f TB/s, median %.2f; sweep-to-sweep difference of one triple: median %.3f, max %.3f" % (
    len(r), r.min(), r.max(), np.median(r), np.median(d), d.max()))
per = [np.mean([np.mean(v) for t, v in res.items() if i in t]) for i in range(K)]
print("mean rate of the triples containing buffer i:", ["%.2f" % p for p in per])
order = sorted(res.items(), key=lambda kv: -np.mean(kv[1]))
print("best :", [(t, "%.2f" % np.mean(v)) for t, v in order[:4]])
print("worst:", [(t, "%.2f" % np.mean(v)) for t, v in order[-4:]])
print("mean rate of the triples containing buffers i and j:")
for i in range(K):
    print("  " + " ".join("%5.2f" % np.mean([np.mean(v) for t, v in res.items() if i in t and j in t]) if i != j else "  -  " for j in range(K)))
print("triples of three consecutive allocations:", [(t, "%.2f" % np.mean(res[t])) for t in [(i, i + 1, i + 2) for i in range(K - 2)]])
