#!/usr/bin/env python3
"""Time one meta-fine-tuning training step (train.py --fine_tune: GnnNet.set_forward_loss_finetune + backward + outer Adam,
gnnnet.py:106-231) the way MetaTemplate.train_loop_finetune runs it.   python tools/metafinetune_time.py [steps] [n_shot: 5 | 20 | 50]
MFT_TRAIN_GRAPH=0 / MFT_ADAPT_GRAPH=0 / MFT_ADAPT_BATCHED_TRUNK=0 switch the round-3 pieces off one by one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import graph_step, optim, synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_shot = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if n_shot == 50:
    from meta_fine_tuning_amd.methods import gnnnet_copy
    GnnNet = gnnnet_copy.GnnNet
model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=n_shot).cuda()
model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
model.train(); model.n_query = 16
opt = optim.Adam(model.parameters())
eps = [synthetic.train_episode(100 + i, 5, n_shot, 16, 84).cuda() for i in range(4)]
np.random.seed(10)
graphed = graph_step.for_loop(model, model.set_forward_loss_finetune)


def step(i):
    if graphed is not None:
        loss = graphed(eps[i % 4])
    else:
        opt.zero_grad()
        loss = model.set_forward_loss_finetune(eps[i % 4])
        loss.backward()
    opt.step()
    return loss


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
print("%d-shot meta-fine-tuning step: %.1f ms per episode (outer half graphed: %s)" % (n_shot, (time.perf_counter() - t0) / steps * 1e3,
                                                                            graphed is not None and graphed.graph is not None))
