#!/usr/bin/env python3
"""Per-launch breakdown of engine.adapt_last_block (the inner loop of ONE meta-fine-tuning training episode, E = 1):
python tools/adapt_breakdown.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, backbone, engine as eng, synthetic
from step_breakdown import Proxy

px = Proxy(_lib.lib())
_lib._lib = px
eng.ADAPT_GRAPH = False
mod = backbone.ResNet10().cuda()
mod.load_state_dict(synthetic.resnet10_state_dict(seed=41))
mod.train()
x_a = synthetic.train_episode(77, 5, 5, 16, 84)[:, :5].reshape(25, 3, 84, 84).cuda()
y_a = np.repeat(range(5), 5).astype(np.int32)
eng.adapt_last_block(mod, x_a, y_a, epochs=15, batch_size=4)
torch.cuda.synchronize()
px.on = True
eng.adapt_last_block(mod, x_a, y_a, epochs=15, batch_size=4)
torch.cuda.synchronize()
px.on = False
agg = collections.OrderedDict()
for k, a, b in px.rec:
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print("== adapt_last_block: %d launches, %.2f ms inside them ==" % (len(px.rec), tot))
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-100s %4d x %8.1f us = %7.2f ms" % (k[:100], n, ms / n * 1e3, ms))
