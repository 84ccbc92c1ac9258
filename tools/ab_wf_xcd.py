"""Fused weight gradient + Adam + next forward launches alone, with / without the one-XCD-per-episode workgroup order.
    gpurun -- python tools/ab_wf_xcd.py [E]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_fine_tuning_amd  # noqa: F401,E402
from meta_fine_tuning_amd import _lib, ops  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
lib = _lib.lib()
gen = torch.Generator(device=dev)
gen.manual_seed(1)
for name, Cin, k, stride, pad, H, mode in (("C2", 512, 3, 1, 1, 3, ops.WF_RAW), ("C1", 256, 3, 2, 1, 6, ops.WF_RAW), ("sc", 256, 1, 2, 0, 6, ops.WF_RAW)):
    Cout = 512
    OH = (H + 2 * pad - k) // stride + 1
    n = E * 5
    x = torch.randn(n, H, H, Cin, device=dev, generator=gen)
    xn = torch.randn(n, H, H, Cin, device=dev, generator=gen)
    dy = torch.randn(n, OH, OH, Cout, device=dev, generator=gen) * 1e-3
    w = torch.randn(E, Cout, k * k * Cin, device=dev, generator=gen) * 0.02
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    raw = torch.empty(n * OH * OH, Cout, device=dev)
    for on in (1, 0, 1, 0):
        lib.mft_wgrad_fwd_set_xcd(on)
        ops.wgrad_adam_next_forward(x, dy, w, m, v, k, k, stride, pad, 1, 5, x_next=xn, mode=mode, raw=raw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for it in range(10):
            ops.wgrad_adam_next_forward(x, dy, w, m, v, k, k, stride, pad, 2 + it, 5, x_next=xn, mode=mode, raw=raw)
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) * 100.0
        print("%-3s E=%d xcd=%d  %8.1f us  %6.2f TB/s (w, m, v bytes)" % (name, E, on, us, 24.0 * w.numel() / us / 1e6))
lib.mft_wgrad_fwd_set_xcd(1)
