#!/usr/bin/env python3
"""ResNet10 forward throughput (train-mode BN, groups of 100 images as in the transductive final pass) -- the north_star's
'MFMA utilisation on ResNet10 forward' figure.  Usage: forward_tflops.py [n_images] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import functional as Fn, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
size = int(sys.argv[2]) if len(sys.argv) > 2 else 84
F_IMG = {84: 0.28585e9, 224: 1.7774e9}[size]
dev = "cuda:0"
sd = synthetic.resnet10_state_dict(seed=0)
x = torch.randn(n, size, size, 3, device=dev)
which = sys.argv[3] if len(sys.argv) > 3 else "both"            # fp32 | x3 | both
for x3 in {"fp32": (False,), "x3": (True,), "both": (False, True)}[which]:
    W = Fn.ResNet10Weights(sd, dev, x3=x3)
    arena = Fn.Arena(dev)
    Fn.resnet10_forward(W, x, arena, ipg=100, tag="f")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        Fn.resnet10_forward(W, x, arena, ipg=100, tag="f")
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    tf = n * F_IMG / ms / 1e9
    print("ResNet10 forward %dx%d, %d images, trunk.4-6 on %s: %.2f ms = %.1f TFLOP/s algorithmic = %.2f of the fp32-MFMA peak (157.3)"
          % (size, size, n, ("f16x2" if W.f16x2 else "bf16x3") if x3 else "fp32 MFMA", ms, tf, tf / 157.3))
