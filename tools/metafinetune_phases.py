#!/usr/bin/env python3
"""Where one meta-fine-tuning training step (train.py --fine_tune: gnnnet.py:106-231) spends its wall time: synchronised marks
around the phases of GnnNet.set_forward_finetune + backward + outer Adam.   python tools/metafinetune_phases.py [steps]"""
import collections
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, optim, synthetic
from meta_fine_tuning_amd import autograd_ops as AG
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods import gnnnet as G

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
model = G.GnnNet(model_dict["ResNet10"], n_way=5, n_support=5).cuda()
model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
model.train(); model.n_query = 16
opt = optim.Adam(model.parameters())
eps = [synthetic.train_episode(100 + i, 5, 5, 16, 84).cuda() for i in range(4)]
np.random.seed(10)
acc = collections.OrderedDict()


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


eng.adapt_last_block = timed("adapt_last_block (inner loop, 105 Adam steps)", eng.adapt_last_block)
G.eng.adapt_last_block = eng.adapt_last_block
copy.deepcopy = timed("copy.deepcopy (theta_pre, theta_adapted)", copy.deepcopy)
G.GnnNet.MAML_update = timed("MAML_update", G.GnnNet.MAML_update)
AG.gnnnet_head = timed("fc + GNN head forward", AG.gnnnet_head)


def step(i):
    opt.zero_grad()
    loss = model.set_forward_loss_finetune(eps[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); acc["backward"] = acc.get("backward", 0.0) + time.perf_counter() - t0
    t0 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); acc["outer Adam"] = acc.get("outer Adam", 0.0) + time.perf_counter() - t0


step(0); step(1)
acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / steps
print("meta-fine-tuning step (with the phase syncs): %.1f ms per episode" % (tot * 1e3))
for k, v in acc.items():
    print("  %-52s %7.2f ms" % (k, v / steps * 1e3))
print("  %-52s %7.2f ms" % ("everything else (state-dict copies, feature forwards, ...)", (tot - sum(acc.values()) / steps) * 1e3))
