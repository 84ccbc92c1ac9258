#!/usr/bin/env python3
"""Upper bound of folding the trunk's three BatchNorm1-apply launches into the following convolution's loader: the inner loop timed with
those launches simply dropped (results wrong -- timing only), alternated with the full step in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meta_fine_tuning_amd
from meta_fine_tuning_amd import engine as eng, ops, synthetic
E=128; dev="cuda:0"
e = eng.FinetuneEngine(synthetic.gnnnet_state_dict(seed=0), 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E, device=dev)
ep = synthetic.test_episode_device(1, dev)
for s in range(E): e.load_episode(s, ep)
e.adapt.reset(e.W); e.prepare_batch()
rs = np.random.RandomState(0)
tables = e.step_tables([[rs.permutation(500)] for _ in range(E)], E)
tables = (tables*3)[:300]
orig = ops.bn_apply
def skip_r1(x2d, C, rpg, ng, mean, rstd, g, b, act=0, res=None, res_bn=None, out=None, **kw):
    if res is None: return out
    return orig(x2d, C, rpg, ng, mean, rstd, g, b, act=act, res=res, res_bn=res_bn, out=out, **kw)
e.inner_loop(tables[:20]); torch.cuda.synchronize()
for alt in range(3):
    for tag, fn in (("full", orig), ("without the three BN1-apply launches (upper bound of the fold)", skip_r1)):
        ops.bn_apply = fn
        e.inner_loop(tables[:10]); torch.cuda.synchronize()
        t0=time.time(); e.inner_loop(tables); torch.cuda.synchronize()
        print("%-70s %.3f ms/step" % (tag, (time.time()-t0)/len(tables)*1e3))
