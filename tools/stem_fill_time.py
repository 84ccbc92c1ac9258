#!/usr/bin/env python3
"""Stem-cache fill of one lockstep batch (E = 128 x 500 images of 84x84): ONE fused launch (csrc/stem.hip: stem_cache_kernel) vs the three
launches through a full-resolution buffer.  Usage: python tools/stem_fill_time.py [n_images]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import functional as Fn, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
dev = "cuda:0"
W = Fn.ResNet10Weights(synthetic.resnet10_state_dict(seed=0), dev)
x = torch.randn(n, 84, 84, 3, device=dev)
for fused in ("1", "0"):
    os.environ["MFT_STEM_FUSED_FILL"] = fused
    c = Fn.StemCache(W, n, 84, dev, pooled=True)
    c.fill(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        c.fill(x)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print("%-60s %8.2f ms per %d images  (%.1f TFLOP/s of stem convolution; cache %.1f GB)" % (
        "ONE launch (conv + moments + window min/max)" if c.fused_fill else "three launches through a full-resolution buffer",
        ms, n, n * 33.2e6 / ms / 1e9, c.nbytes() / 1e9))
    del c
