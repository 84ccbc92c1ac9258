#!/usr/bin/env python3
"""Episodes/s at small episode batches: eager single stream vs two-stream pipeline vs hipGraph replay of the inner step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, synthetic

dev = "cuda:0"
state = synthetic.gnnnet_state_dict(seed=0)
base = [synthetic.test_episode_device(i, dev) for i in range(4)]
for E in (1, 4, 16, 32):
    pool = [base[i % 4] for i in range(E)]
    row = []
    for name, kw in (("eager", dict(pipeline=False)), ("2-stream", dict(pipeline=True)), ("hipGraph", dict(graph=True))):
        e = eng.FinetuneEngine(state, 5, 5, 15, 84, n_views=19, fine_tune_epoch=5, episodes_per_batch=E, device=dev, **kw)
        e.run_batch(pool); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run_batch(pool); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row.append("%s %.2f eps/s (%.2f ms/step)" % (name, E / dt, dt / 500 * 1e3))
        del e
        torch.cuda.empty_cache()
    print("E=%-3d %s" % (E, " | ".join(row)))
