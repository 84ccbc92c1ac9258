#!/usr/bin/env python3
"""What does each part of the frozen-trunk stream cost the lockstep step in WALL time (it overlaps the HBM-bound last-block
stream, so its standalone time says little)?  Re-times the inner loop with parts of the trunk's launches suppressed
(results are then wrong -- timing only).   Usage: trunk_cost.py [E] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, ops, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = "cuda:0"
e = eng.FinetuneEngine(synthetic.gnnnet_state_dict(seed=0), 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E, device=dev)
ep = synthetic.test_episode_device(1, dev)
for s in range(E):
    e.load_episode(s, ep)
e.adapt.reset(e.W)
e.prepare_batch()
rs = np.random.RandomState(0)
perms = [[rs.permutation(500)] for _ in range(E)]
tables = e.step_tables(perms, E)[:steps]

orig = {k: getattr(ops, k) for k in ("conv2d_x3_bnstats", "conv2d_x3", "bn_apply", "bn_combine_moments")}
orig_gather = e.stem.gather


def run(tag, skip):
    for k, f in orig.items():
        setattr(ops, k, f)
    e.stem.gather = orig_gather
    e.inner_loop(tables[:10]); torch.cuda.synchronize()          # everything has run once: buffers exist
    if "x3" in skip:
        ops.conv2d_x3_bnstats = lambda x, w3, Cout, KH, KW, stride, pad, ipg, out, ws, mean, rstd, **kw: (out, mean, rstd)
        ops.conv2d_x3 = lambda x, w3, Cout, KH, KW, stride, pad, out=None: out
    if "bn" in skip:
        ops.bn_apply = lambda x2d, C, rpg, ng, mean, rstd, g, b, act=0, res=None, res_bn=None, out=None, **kw: out
    if "gather" in skip:
        ops.bn_combine_moments = lambda *a, **kw: None
        e.stem.gather = lambda idx, n, m, s, g, b, ipg, out, planes=None: out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.inner_loop(tables)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(tables) * 1e3
    print("%-44s %.3f ms per lockstep step" % (tag, ms))
    return ms


base = run("full step", ())
a = run("without the trunk convolutions (conv_x3)", ("x3",))
b = run("without BatchNorm-apply launches", ("bn",))
c = run("without stem gather + moment combine", ("gather",))
d = run("without any trunk launch", ("x3", "bn", "gather"))
run("full step (again)", ())
print("wall cost: conv_x3 %.3f ms, BN-apply %.3f, gather %.3f; all trunk %.3f" % (base - a, base - b, base - c, base - d))
