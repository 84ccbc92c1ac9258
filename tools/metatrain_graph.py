#!/usr/bin/env python3
"""Meta-training step (BASELINE configs[3]) with set_forward_loss + backward captured in ONE hipGraph (the outer Adam and, under
data parallelism, the gradient all-reduce stay outside): python tools/metatrain_graph.py [steps] [n_shot: 5 | 50 (gnnnet_copy)]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import optim, synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_shot = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if n_shot == 50:
    from meta_fine_tuning_amd.methods import gnnnet_copy
    GnnNet = gnnnet_copy.GnnNet


def build():
    torch.manual_seed(0)
    model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=n_shot).cuda()
    model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
    model.train()
    model.n_query = 16
    return model, optim.Adam(model.parameters())


eps = [synthetic.train_episode(5000 + i, 5, n_shot, 16, 84).cuda() for i in range(4)]

# eager
model, opt = build()
losses_e = []
for i in range(3 + steps):
    if i == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    opt.zero_grad()
    loss = model.set_forward_loss(eps[i % 4])
    loss.backward()
    opt.step()
    losses_e.append(loss.detach())
torch.cuda.synchronize()
dt_e = (time.perf_counter() - t0) / steps
losses_e = [float(l) for l in losses_e]

# graphed
model, opt = build()
static_x = eps[0].clone()
losses_g = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        static_x.copy_(eps[i % 4])
        loss = model.set_forward_loss(static_x)
        loss.backward()
        opt.step()
        losses_g.append(loss.detach().clone())
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = model.set_forward_loss(static_x)
    static_loss.backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + steps):
    static_x.copy_(eps[i % 4])
    g.replay()
    opt.step()
    losses_g.append(static_loss.detach().clone())
torch.cuda.synchronize()
dt_g = (time.perf_counter() - t0) / steps
losses_g = [float(l) for l in losses_g]
print("eager   %.2f ms per step" % (dt_e * 1e3))
print("graphed %.2f ms per step" % (dt_g * 1e3))
print("max |loss difference| over %d steps: %.3e" % (len(losses_e), max(abs(a - b) for a, b in zip(losses_e, losses_g))))
print("eager  ", " ".join("%.5f" % l for l in losses_e[:8]))
print("graphed", " ".join("%.5f" % l for l in losses_g[:8]))
