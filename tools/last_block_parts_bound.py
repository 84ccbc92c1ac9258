#!/usr/bin/env python3
"""What each launch of the last-block stream costs the lockstep step: the inner loop timed with one launch at a time replaced by a
no-op (results wrong -- timing only), alternated with the full step in one process."""
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
lib = ops._lib.lib()
names = ("mft_block_entry_small_forward", "mft_block_exit_small_forward", "mft_ce_pool_bn_backward2", "mft_conv2d_dgrad_bn_backward_small")
orig = {n: getattr(lib, n) for n in names}
orig_wgrad = ops.conv2d_wgrad_adam


def wgrad_without(shape_k):
    def f(xin, dy, w, m, v, Cout, KH, KW, stride, pad, step, **kw):
        if (KH, stride) == shape_k:
            return None
        return orig_wgrad(xin, dy, w, m, v, Cout, KH, KW, stride, pad, step, **kw)
    return f


cases = [("full", None, None)] + [("without " + n, n, None) for n in names] + [
    ("without weight gradient + Adam of trunk.7.C2", None, (3, 1)), ("without weight gradient + Adam of trunk.7.C1", None, (3, 2)),
    ("without weight gradient + Adam of the shortcut", None, (1, 2))]
e.inner_loop(tables[:20]); torch.cuda.synchronize()
base = None
for tag, drop, wk in cases:
    for n in names:
        setattr(lib, n, orig[n])
    ops.conv2d_wgrad_adam = orig_wgrad
    if drop is not None:
        setattr(lib, drop, lambda *a: 0)
    if wk is not None:
        ops.conv2d_wgrad_adam = wgrad_without(wk)
    e.inner_loop(tables[:10]); torch.cuda.synchronize()
    t0 = time.time(); e.inner_loop(tables); torch.cuda.synchronize()
    ms = (time.time() - t0) / len(tables) * 1e3
    base = ms if base is None else base
    print("%-58s %.3f ms/step  (%+.1f %%)" % (tag, ms, (ms - base) / base * 100.0))
for n in names:
    setattr(lib, n, orig[n])
ops.conv2d_wgrad_adam = orig_wgrad
