#!/usr/bin/env python3
"""Standalone time of the dominant launch on trunk.7's three layers at E episodes: wgrad_adam_rows_kernel (gradient + Adam only)
against wgrad_adam_fwd_kernel (the same + the next step's convolution and block epilogue).   python tools/wgrad_fwd_time.py [E]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


if True:
  for name, Cin, k, stride, pad, H, mode in (("C2", 512, 3, 1, 1, 3, ops.WF_EXIT), ("C1", 256, 3, 2, 1, 6, ops.WF_ENTRY), ("shortcut", 256, 1, 2, 0, 6, ops.WF_RAW)):
      n, Cout = E * 5, 512
      OH = (H + 2 * pad - k) // stride + 1
      x = torch.randn(n, H, H, Cin, device=dev, generator=g)
      dy = torch.randn(n, OH, OH, Cout, device=dev, generator=g) * 1e-3
      w = torch.randn(E, Cout, k * k * Cin, device=dev, generator=g) * 0.02
      m, v = torch.zeros_like(w), torch.zeros_like(w)
      raw, act, sc = (torch.randn(n * OH * OH, Cout, device=dev, generator=g) for _ in range(3))
      st = [torch.empty(E, Cout, device=dev) for _ in range(4)]
      gb = torch.ones(E, Cout, device=dev)
      pooled = torch.empty(n, Cout, device=dev)
      by = 24.0 * w.numel()
      t0 = timeit(lambda: ops.conv2d_wgrad_adam(x, dy, w, m, v, Cout, k, k, stride, pad, 3, imgs_per_group=5))
      t1 = timeit(lambda: ops.wgrad_adam_next_forward(x, dy, w, m, v, k, k, stride, pad, 3, 5))
      t2 = timeit(lambda: ops.wgrad_adam_next_forward(x, dy, w, m, v, k, k, stride, pad, 3, 5, x_next=x, mode=mode, raw=raw, act=act, gamma=gb, beta=gb,
                                                      gbs=Cout, mean=st[0], rstd=st[1], sc_raw=sc, gamma_s=gb, beta_s=gb, mean_s=st[2], rstd_s=st[3],
                                                      pooled=pooled))
      print("%-9s E=%d  rows kernel %7.1f us (%.2f TB/s) | walk, update only %7.1f us (%.2f TB/s) | walk + next forward %7.1f us (%.2f TB/s)"
              % (name, E, t0, by / t0 / 1e6, t1, by / t1 / 1e6, t2, by / t2 / 1e6))

