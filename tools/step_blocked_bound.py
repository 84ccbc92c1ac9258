#!/usr/bin/env python3
"""Round-5 go/no-go, step B's bound (VERDICT r04 "next 1"): what would ONE chunk-step of a step-blocked last-block loop cost
with the kernels that exist?  A chunk of c = 3..6 episodes (132-264 MB of w / m / v: the set that could stay in the 256 MiB
Infinity Cache) runs its 500 inner steps back to back, the frozen trunk taken out (MFT_DEBUG_SKIP_TRUNK=1: x6 of a step is a
cached buffer, as it would be with the trunk precomputed for all steps), each step replayed from ONE hipGraph (no host launch
cost).  The step-blocked schedule of a 128-episode batch then costs (128 / c) x 500 x (this chunk-step) + the trunk pass;
today's lockstep schedule costs 500 x 2.89 ms.  Usage: python tools/step_blocked_bound.py"""
import os
import sys

os.environ["MFT_DEBUG_SKIP_TRUNK"] = "1"
os.environ.setdefault("MFT_SLAB_CANDIDATES", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, synthetic

dev = "cuda:0"
sd = synthetic.gnnnet_state_dict(seed=0)
print("# chunk-step of the last-block loop alone (trunk skipped), 500 steps, E = c episodes; us per chunk-step")
print("# %3s %12s %14s %14s %22s" % ("c", "w/m/v MB", "graph replay", "eager launches", "=> 128-episode batch s"))
for c in (1, 2, 3, 4, 5, 6, 8, 16):
    eps = [synthetic.test_episode_device(100 + i, dev, 5, 5, 15, 84, gen_examples=17) for i in range(c)]
    res = {}
    for graph in (True, False):
        e = eng.FinetuneEngine(sd, n_views=19, fine_tune_epoch=5, episodes_per_batch=c, device=dev, graph=graph, pipeline=False,
                               fuse_next=False)
        e._ingest(eps, False)
        e.prepare_batch()
        rs = np.random.RandomState(1)
        perms = [[rs.permutation(500) for _ in range(5)] for _ in range(c)]
        tables = e.step_tables(perms, c)
        ts = []
        for rep in range(3):
            e.adapt.reset(e.W)
            e.step_dev.zero_()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            e.inner_loop(tables)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / len(tables))
        res[graph] = min(ts[1:])
        e.close()
        del e
    best = min(res.values())
    print("  %3d %12.1f %14.1f %14.1f %22.3f" % (c, c * 44.08, res[True], res[False], 128.0 / c * 500 * best * 1e-6))
    del eps
    torch.cuda.empty_cache()
print("# today's lockstep schedule at E = 128: 500 x 2.89 ms = 1.445 s per batch INCLUDING the trunk (0.65 ms per step alone)")
