#!/usr/bin/env python3
"""Wall time of each phase of FinetuneEngine.run_batch (synchronised between phases).  Usage: phase_times.py [E]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pipe = (len(sys.argv) <= 2) or sys.argv[2] != "nopipe"
dev = "cuda:0"
if os.environ.get("MFT_DEBUG_TILE"):
    from meta_fine_tuning_amd import _lib
    for t in os.environ["MFT_DEBUG_TILE"].split(","):
        _lib.lib().mft_debug_set_conv_tile(int(t))
if os.environ.get("MFT_X3_TILE"):
    from meta_fine_tuning_amd import _lib
    for t in os.environ["MFT_X3_TILE"].split(","):
        _lib.lib().mft_debug_set_x3_tile(int(t))
state = synthetic.gnnnet_state_dict(seed=0)
e = eng.FinetuneEngine(state, 5, 5, 15, 84, n_views=19, fine_tune_epoch=5, episodes_per_batch=E, device=dev, pipeline=pipe)
pool = [synthetic.test_episode_device(i, dev) for i in range(min(E, 8))]
pool = [pool[i % len(pool)] for i in range(E)]
e.run_batch(pool)
torch.cuda.synchronize()


def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


for rep in range(2):
    _, t_load = t(lambda: [e.load_episode(s, pool[s]) for s in range(E)])
    _, t_reset = t(lambda: e.adapt.reset(e.W))
    _, t_prep = t(e.prepare_batch)
    perms, t_perm = t(lambda: [eng.draw_perms(e.n_total, e.epochs) for _ in range(E)])
    tables, t_tab = t(lambda: e.step_tables(perms, E))
    _, t_inner = t(lambda: e.inner_loop(tables))
    _, t_fin = t(e.final_scores)
    tot = t_load + t_reset + t_prep + t_perm + t_tab + t_inner + t_fin
    print("E=%d pipeline=%s: load %.1f | reset %.1f | stem cache %.1f | perms %.1f | tables %.1f | inner loop %.1f (%.3f ms/step) | "
          "final+GNN %.1f | total %.1f ms -> %.1f episodes/s" % (E, pipe, t_load, t_reset, t_prep, t_perm, t_tab, t_inner,
                                                               t_inner / len(tables), t_fin, tot, E / tot * 1e3))
