#!/usr/bin/env python3
"""For the GNN's 1x1 layers: the data gradient dh = dz @ W as (a) the BT form of the implicit-GEMM kernel reading the forward pack
(what the backward launches today) and (b) the forward GEMM kernel on a pre-transposed pack.  Replayed from a hipGraph of 20 back-to-back
dependent launches each (the regime of the meta-training step).  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge

ge.build()
from meta_fine_tuning_amd import ops

dev = "cuda"
for rows, cin, cout in ((7440, 192, 192), (7440, 96, 192), (7440, 192, 96), (7440, 96, 96), (29760, 192, 192), (480, 288, 64), (105, 512, 128)):
    g = torch.Generator(device=dev).manual_seed(1)
    dz = torch.randn(rows, cout, device=dev, generator=g)
    w = torch.randn(cout, cin, device=dev, generator=g) * 0.05
    wpk = ops.pack_conv_weight(w)                     # [cout, cin]
    wt = ops.pack_conv_weight(w.t().contiguous())     # [cin, cout]: the forward pack of W^T
    a = ops.conv2d_dgrad(dz.view(rows, 1, 1, cout), wpk, cin, 1, 1, 0).view(rows, cin)
    b = ops.gemm(dz, cout, wt, cin)
    err = float((a - b).abs().max()) / float(a.abs().max())
    res = []
    for fn in (lambda x: ops.conv2d_dgrad(x.view(rows, 1, 1, cout), wpk, cin, 1, 1, 0).view(rows, cin), lambda x: ops.gemm(x, cout, wt, cin)):
        if cin != cout:                               # chain needs square shapes: time independent launches instead
            x = dz
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    fn(x)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    for _ in range(20):
                        y = fn(x)
        else:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                x = dz
                for _ in range(3):
                    x = fn(x) * 1.0
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    x = dz
                    for _ in range(20):
                        x = fn(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 200)
    print("rows %6d cin %3d cout %3d:  dgrad (BT) %6.2f us   forward GEMM on W^T %6.2f us   (max rel diff %.1e)" % (rows, cin, cout, res[0], res[1], err), flush=True)
