#!/usr/bin/env python3
"""GNN head (fc + 3 x Wcompute + 3 x Gconv) time per batch of E episodes: fused pair-MLP kernels (csrc/pair_mlp.hip) vs the
materialised round-1 sequence.  Usage: gnn_time.py N_SUPPORT E [fused|unfused|both] [reps]
Prints ms per call, algorithmic TFLOP/s of the per-pair MLP (596,160 FLOP per (b,i,j) pair over the three
Wcomputes, all N*N pairs counted -- the fused form executes half of them) and the arena bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import functional as Fn, synthetic

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
which = sys.argv[3] if len(sys.argv) > 3 else "both"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fold = ns == 50
gs = ns // 2 if fold else ns
dev = "cuda:0"
G = Fn.GnnHeadWeights(synthetic.gnn_head_state_dict(seed=5), dev, 5)
feats = torch.randn(E * 5 * (ns + 15), 512, device=dev)
N = 5 * (gs + 1)
flop = 596160.0 * E * 15 * N * N
for mode in (("fused", "unfused") if which == "both" else (which,)):
    Fn.FUSED_PAIR_MLP = mode == "fused"
    arena = Fn.Arena(dev)
    try:
        sc = Fn.gnnnet_scores(G, feats, E, 5, gs, 15, arena, fold=fold)
    except torch.OutOfMemoryError as ex:
        print("%-8s E=%d N=%d: out of memory (%s)" % (mode, E, N, str(ex)[:60]))
        continue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sc = Fn.gnnnet_scores(G, feats, E, 5, gs, 15, arena, fold=fold)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%-8s E=%d N=%d: %.2f ms per batch, %.1f algorithmic TFLOP/s, arena %.2f GB, checksum %.6f"
          % (mode, E, N, ms, flop / ms / 1e9, arena.nbytes() / 2 ** 30, float(sc.double().sum())))
    del arena
    torch.cuda.empty_cache()
