#!/usr/bin/env python3
"""Run each trunk x3 convolution a few times (for rocprofv3 --pmc FETCH_SIZE): usage x3_fetch.py [E] [xcd 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops, _lib
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
_lib.lib().mft_debug_set_x3_tile(20 + int(sys.argv[2]) if len(sys.argv) > 2 else 21)
n = E * 5
for (cin, cout, k, s, p, H) in [(64, 64, 3, 1, 1, 21), (128, 128, 3, 1, 1, 11), (256, 256, 3, 1, 1, 6), (128, 256, 3, 2, 1, 11)]:
    x = torch.randn(n, H, H, cin, device="cuda")
    w3 = ops.split_weight_x3(ops.pack_conv_weight(torch.randn(cout, cin, k, k, device="cuda") * 0.05))
    for _ in range(3):
        ops.conv2d_x3(x, w3, cout, k, k, s, p)
    torch.cuda.synchronize()
    print("in %.0f MB out %.0f MB" % (x.numel() * 4 / 1e6, n * ((H + 2 * p - k) // s + 1) ** 2 * cout * 4 / 1e6))
