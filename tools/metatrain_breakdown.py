#!/usr/bin/env python3
"""Per-launch breakdown of one meta-training step (set_forward_loss + backward + outer Adam, meta_template.py:76-92) at BASELINE
configs[3] (5-way 5-shot, 16 queries, 84x84): HIP events around every C-ABI launch, aggregated by (entry point, shape).
    python tools/metatrain_breakdown.py [steps]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, optim, synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet
from step_breakdown import Proxy

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def main():
    px = Proxy(_lib.lib())
    _lib._lib = px
    torch.manual_seed(0)
    model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=5).cuda()
    model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
    model.train()
    model.n_query = 16
    opt = optim.Adam(model.parameters())
    eps = [synthetic.train_episode(5000 + i, 5, 5, 16, 84).cuda() for i in range(2)]

    def step(i):
        opt.zero_grad()
        loss = model.set_forward_loss(eps[i % 2])
        loss.backward()
        opt.step()
        return loss
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / steps
    px.on = True
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    px.on = False
    agg = collections.OrderedDict()
    for k, a, b in px.rec:
        agg.setdefault(k, [0, 0.0])
        agg[k][0] += 1
        agg[k][1] += a.elapsed_time(b)
    tot = sum(v[1] for v in agg.values()) / steps
    print("== meta-training step: %.2f ms wall (no events); %d C-ABI launches per step, %.2f ms inside them ==" % (plain * 1e3, len(px.rec) // steps, tot))
    by_name = collections.OrderedDict()
    for k, (n, ms) in agg.items():
        nm = k.split("(")[0]
        by_name.setdefault(nm, [0, 0.0])
        by_name[nm][0] += n
        by_name[nm][1] += ms
    print("-- by entry point")
    for nm, (n, ms) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
        print("%-44s %5.1f/step %8.1f us/step %5.1f%%" % (nm, n / steps, ms / steps * 1e3, 100 * ms / steps / tot))
    print("-- by shape (top 60)")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%-104s %4.1f/step %8.1f us each" % (k[:104], n / steps, ms / n * 1e3))


if __name__ == "__main__":
    main()
