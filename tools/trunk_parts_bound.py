#!/usr/bin/env python3
"""Upper bounds for further trunk work: the inner loop timed with groups of trunk convolutions dropped (results wrong -- timing only),
alternated with the full step in one process.  What a launch costs the step is what could at most be won by improving it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, ops, synthetic
E = 128; dev = "cuda:0"
e = eng.FinetuneEngine(synthetic.gnnnet_state_dict(seed=0), 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E, device=dev)
ep = synthetic.test_episode_device(1, dev)
for s in range(E):
    e.load_episode(s, ep)
e.adapt.reset(e.W); e.prepare_batch()
rs = np.random.RandomState(0)
tables = (e.step_tables([[rs.permutation(500)] for _ in range(E)], E) * 3)[:300]
orig_bn, orig_x3 = ops.conv2d_x3_bnstats, ops.conv2d_x3


def make(pred):
    def f(x, w3, Cout, KH, KW, stride, pad, ipg, out, ws, mean, rstd, **kw):
        if pred(KH, stride):
            return out, mean, rstd
        return orig_bn(x, w3, Cout, KH, KW, stride, pad, ipg, out, ws, mean, rstd, **kw)
    return f


cases = (("full", None), ("without the 1x1 shortcut convolutions", lambda k, s: k == 1),
         ("without the 3x3 / stride-2 convolutions", lambda k, s: k == 3 and s == 2),
         ("without the 3x3 / stride-1 convolutions", lambda k, s: k == 3 and s == 1))
e.inner_loop(tables[:20]); torch.cuda.synchronize()
for alt in range(2):
    for tag, pred in cases:
        ops.conv2d_x3_bnstats = orig_bn if pred is None else make(pred)
        e.inner_loop(tables[:10]); torch.cuda.synchronize()
        t0 = time.time(); e.inner_loop(tables); torch.cuda.synchronize()
        print("%-48s %.3f ms/step" % (tag, (time.time() - t0) / len(tables) * 1e3))
ops.conv2d_x3_bnstats = orig_bn
