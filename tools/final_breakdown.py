#!/usr/bin/env python3
"""Per-launch breakdown of FinetuneEngine.final_scores (100-image transductive pass + GNN head) at E episodes."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine as eng, synthetic
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_breakdown import Proxy

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
real = _lib.lib()
px = Proxy(real)
_lib._lib = px
dev = "cuda:0"
e = eng.FinetuneEngine(synthetic.gnnnet_state_dict(seed=0), 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E,
                       device=dev, pipeline=False)
ep = synthetic.test_episode_device(1, dev)
for s in range(E):
    e.load_episode(s, ep)
e.adapt.reset(e.W)
e.final_scores(); torch.cuda.synchronize()
px.on = True
e.final_scores(); torch.cuda.synchronize()
px.on = False
agg = collections.OrderedDict()
for k, a, b in px.rec:
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print("final_scores E=%d: %.1f ms of launches" % (E, tot))
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-100s %3d %9.1f us %5.1f%%" % (k[:100], n, ms * 1e3, 100 * ms / tot))
