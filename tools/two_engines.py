#!/usr/bin/env python3
"""Experiment: one engine with E episodes vs L engines with E/L episodes each running concurrently on their own stream
sets (do the dependent-launch gaps of one lockstep chain get filled by the other chain?).  Usage: two_engines.py [E] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda:0"
state = synthetic.gnnnet_state_dict(seed=0)
base = [synthetic.test_episode_device(i, dev) for i in range(8)]
pool = [base[i % 8] for i in range(E)]
for lanes in (1, L, 1, L):
    per = E // lanes
    engines = [eng.FinetuneEngine(state, 5, 5, 15, 84, n_views=19, fine_tune_epoch=5, episodes_per_batch=per, device=dev)
               for _ in range(lanes)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]

    def run():
        outs = []
        for i, (e, s) in enumerate(zip(engines, streams)):
            with torch.cuda.stream(s):
                outs.append(e.run_batch(pool[i * per:(i + 1) * per]))
        return outs
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print("%d engine(s) x %d episodes: %.1f ms per %d episodes -> %.1f episodes/s" % (lanes, per, dt * 1e3, E, E / dt))
    del engines
    torch.cuda.empty_cache()
