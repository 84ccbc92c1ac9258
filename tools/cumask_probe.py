#!/usr/bin/env python3
"""CU-mask experiments on the box: (1) which mask bits land on which XCD, (2) how the HBM-bound wgrad+Adam launch and an
MFMA-bound trunk convolution scale with the number of CUs they may use, (3) what they cost when co-run on
complementary CU sets.  Usage: python tools/cumask_probe.py [E]"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NBITS = 256
lib = _lib.lib()


def make_stream(bits):
    words = (ctypes.c_uint * (NBITS // 32))()
    for b in bits:
        words[b // 32] |= (1 << (b % 32))
    out = ctypes.c_void_p()
    rc = lib.mft_stream_create_cumask(ctypes.cast(words, ctypes.c_void_p), NBITS // 32, ctypes.byref(out))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(out.value)


def probe(stream, n_blocks=4096):
    out = torch.zeros(2 * n_blocks, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        lib.mft_probe_placement(ops._p(out), n_blocks, 20000, ops._stream())
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(-1, 2).astype(np.int64) & 0xffffffff
    xcc = o[:, 0] & 0xf
    hw = o[:, 1]
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cus = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = collections.Counter(x for x, _, _, _ in cus)
    return cus, per_xcc


def timeit(fn, stream, iters=10):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        a.record(stream)
        for _ in range(iters):
            fn()
        b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    print("== mask bit -> XCD mapping ==")
    for name, bits in (("bits 0..31", range(32)), ("bits 0..7", range(8)), ("bits i%8==0", range(0, 256, 8)),
                       ("bits i%32<4", [i for i in range(256) if i % 32 < 4]), ("all 256", range(256))):
        cus, per = probe(make_stream(list(bits)))
        print("%-14s -> %3d distinct CUs; per XCC: %s" % (name, len(cus), dict(sorted(per.items()))))

    # families: A(n) = n CUs per XCD if bits interleave over XCDs (bit i -> XCD i%8); B(n) if blocked (bit i -> XCD i//32)
    def famA(n, lo=0):
        return [i for i in range(256) if lo <= (i // 8) < lo + n]

    def famB(n, lo=0):
        return [i for i in range(256) if lo <= (i % 32) < lo + n]

    n = E * 5
    # HBM-bound: wgrad + Adam of trunk.7.C2 (per-episode weights)
    r1 = torch.randn(n, 3, 3, 512, device="cuda")
    dc2 = torch.randn(n, 3, 3, 512, device="cuda") * 1e-3
    w = torch.randn(E, 512, 4608, device="cuda") * 0.02
    m = torch.zeros_like(w)
    v = torch.zeros_like(w)
    byt = 6.0 * 4 * w.numel()

    def f_wgrad():
        ops.conv2d_wgrad_adam(r1, dc2, w, m, v, 512, 3, 3, 1, 1, 3, imgs_per_group=5)

    # MFMA-bound: trunk.6.C2 (shared weights)
    x6 = torch.randn(n, 6, 6, 256, device="cuda")
    w6 = torch.randn(256, 2304, device="cuda") * 0.02
    o6 = torch.empty(n, 6, 6, 256, device="cuda")
    fl = 2.0 * n * 36 * 256 * 2304

    def f_conv():
        ops.conv2d(x6, w6, 256, 3, 3, 1, 1, out=o6)

    # weight-streaming forward: trunk.7.C2 per-episode
    o7 = torch.empty(n, 3, 3, 512, device="cuda")

    def f_fwd7():
        ops.conv2d(r1, w, 512, 3, 3, 1, 1, imgs_per_group=5, out=o7)

    full = make_stream(list(range(256)))
    print("== scaling with CUs (family A = bits with i//8 < n, family B = bits with i%%32 < n) ==")
    print("full mask: wgrad+adam %.0f us (%.2f TB/s) | conv6 %.0f us (%.1f TF) | fwd7 %.0f us (%.2f TB/s)" % (
        timeit(f_wgrad, full), byt / timeit(f_wgrad, full) / 1e6, timeit(f_conv, full), fl / timeit(f_conv, full) / 1e6,
        timeit(f_fwd7, full), 4.0 * w.numel() / timeit(f_fwd7, full) / 1e6))
    for fam, fname in ((famA, "A"), (famB, "B")):
        for k in (8, 12, 16, 20, 24):
            s = make_stream(fam(k))
            _, per = probe(s)
            t1 = timeit(f_wgrad, s)
            t2 = timeit(f_conv, s)
            t3 = timeit(f_fwd7, s)
            print("fam %s n=%2d (%3d CUs, per XCC %s): wgrad+adam %.0f us (%.2f TB/s) | conv6 %.0f us (%.1f TF) | fwd7 %.0f us (%.2f TB/s)" % (
                fname, k, 8 * k, dict(sorted(per.items())), t1, byt / t1 / 1e6, t2, fl / t2 / 1e6, t3, 4.0 * w.numel() / t3 / 1e6))

    print("== co-run on complementary CU sets: 10 x wgrad+adam || R x conv6, wall time ==")
    for fam, fname in ((famA, "A"), (famB, "B")):
        for k in (8, 12, 16):
            s_mem = make_stream(fam(k))
            s_mm = make_stream(fam(32 - k, lo=k))
            t_mem = timeit(f_wgrad, s_mem)
            t_mm = timeit(f_conv, s_mm)
            reps = max(1, int(round(10 * t_mem / t_mm)))
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            a.record(cur)
            s_mem.wait_stream(cur)
            s_mm.wait_stream(cur)
            with torch.cuda.stream(s_mem):
                for _ in range(10):
                    f_wgrad()
            with torch.cuda.stream(s_mm):
                for _ in range(reps):
                    f_conv()
            cur.wait_stream(s_mem)
            cur.wait_stream(s_mm)
            b.record(cur)
            torch.cuda.synchronize()
            wall = a.elapsed_time(b) * 1e3
            print("fam %s mem=%3d CUs / mm=%3d CUs: alone wgrad %.0f us, conv %.0f us; co-run 10 wgrad || %d conv: wall %.0f us "
                  "(serial on full chip would be %.0f us)" % (fname, 8 * k, 8 * (32 - k), t_mem, t_mm, reps, wall,
                                                               10 * timeit(f_wgrad, full) + reps * timeit(f_conv, full)))
    # plain two-stream co-run for comparison
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    tw, tc = timeit(f_wgrad, full), timeit(f_conv, full)
    reps = max(1, int(round(10 * tw / tc)))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    a.record(cur)
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(10):
            f_wgrad()
    with torch.cuda.stream(s2):
        for _ in range(reps):
            f_conv()
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    b.record(cur)
    torch.cuda.synchronize()
    print("unmasked two streams: 10 wgrad || %d conv: wall %.0f us (serial %.0f us)" % (reps, a.elapsed_time(b) * 1e3,
                                                                                      10 * tw + reps * tc))


if __name__ == "__main__":
    main()
