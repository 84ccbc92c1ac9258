#!/usr/bin/env python3
"""Per-episode bf16x3 kernels (trunk.7 block entry / exit / C2 data gradient) at small episode batches: waves per workgroup
chosen by episode count (knob 9199) vs the default widest workgroups (16 waves / 256 channels).  Usage: small_e_skinny.py [E ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops, _lib

lib = _lib.lib()
C = 512


def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for E in [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64, 128]:
    n = E * 5
    x6 = torch.randn(n, 6, 6, 256, device="cuda")
    w1 = torch.randn(E, C, 2304, device="cuda") * 0.02
    ws = torch.randn(E, C, 256, device="cuda") * 0.05
    w2 = torch.randn(E, C, 4608, device="cuda") * 0.02
    g = torch.ones(E, C, device="cuda"); b = torch.zeros(E, C, device="cuda")
    c1, r1, sc, c2, out, dc1 = (torch.empty(n, 3, 3, C, device="cuda") for _ in range(6))
    dc2 = torch.randn(n, 3, 3, C, device="cuda") * 1e-3
    st = [torch.empty(E, C, device="cuda") for _ in range(8)]
    feat = torch.empty(n, C, device="cuda")

    def entry():
        assert lib.mft_block_entry_small_forward(ops._p(x6), 256, ops._p(w1), C * 2304, ops._p(ws), C * 256, ops._p(c1), ops._p(r1),
                                                 ops._p(sc), n, 6, 6, 256, C, 2, 5, ops._p(g), ops._p(b), C, ops._p(st[0]),
                                                 ops._p(st[1]), 1e-5, ops._stream()) == 0

    def exit_():
        assert lib.mft_block_exit_small_forward(ops._p(r1), ops._p(w2), C * 4608, ops._p(sc), ops._p(c2), ops._p(out), ops._p(feat),
                                                n, 3, 3, C, 5, ops._p(g), ops._p(b), ops._p(g), ops._p(b), C, ops._p(st[2]),
                                                ops._p(st[3]), ops._p(st[4]), ops._p(st[5]), 1e-5, ops._stream()) == 0

    def dgrad():
        assert lib.mft_conv2d_dgrad_bn_backward_small(ops._p(dc2), C, ops._p(w2), ops._p(dc1), C, n, 3, 3, C, C, 3, 3, 1, 5, C * 4608,
                                                      ops._p(c1), ops._p(r1), ops._p(st[0]), ops._p(st[1]), ops._p(g), C,
                                                      ops._p(st[6]), ops._p(st[7]), ops._stream()) == 0

    res = []
    outs = []
    for knob in (9199, 9100):
        lib.mft_debug_set_conv_tile(knob)
        res.append((t(entry), t(exit_), t(dgrad)))
        outs.append((r1.clone(), out.clone(), dc1.clone()))
    lib.mft_debug_set_conv_tile(9100)
    same = all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
    print("E=%3d  auto: entry %4.0f exit %4.0f dgrad %4.0f us | 16-wave workgroups: entry %4.0f exit %4.0f dgrad %4.0f us | identical results: %s"
          % ((E,) + res[0] + res[1] + (same,)))
