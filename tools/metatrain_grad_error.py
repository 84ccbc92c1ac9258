#!/usr/bin/env python3
"""Gradient error of one meta-training step against the float64 oracle, with the split-precision layers on and off
(MFT_TRAIN_X3): relative L2 error per parameter tensor (median / 90th percentile / max over the 104 tensors; tensors whose true
gradient is exactly zero excluded).  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge

ge.build()
from meta_fine_tuning_amd import functional as Fn, synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet
from oracle import mft_oracle as O

for seed, ep in ((7, 21), (9, 23)):
    sd32 = synthetic.gnnnet_state_dict(seed=seed)
    x = synthetic.train_episode(ep, 5, 5, 16, 84)
    sd = O.clone_state(sd32, torch.float64)
    pkeys = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in pkeys:
        sd[k].requires_grad_(True)
    loss, _ = O.meta_train_loss(sd, x.double(), 5, 5)
    ref = dict(zip(pkeys, torch.autograd.grad(loss, [sd[k] for k in pkeys])))
    got = {}
    for on in (True, False):
        Fn.TRAIN_X3 = on
        model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5)
        model.load_state_dict(sd32)
        model = model.cuda().train()
        model.n_query = 16
        l = model.set_forward_loss(x)
        l.backward()
        g = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
        rel = np.array([float((g[k] - ref[k]).norm() / ref[k].norm()) for k in ref if float(ref[k].norm()) > 1e-9])
        relb = np.array([float((g[k] - ref[k]).norm() / ref[k].norm()) for k in ref if float(ref[k].norm()) > 1e-9 and k.startswith("feature.")])
        got[on] = g
        print("seed %d  MFT_TRAIN_X3=%d  loss %.7f (float64 %.7f)  rel. L2 error vs float64 over %d tensors: median %.2e  p90 %.2e  max %.2e;  backbone only: median %.2e max %.2e"
              % (seed, on, float(l), float(loss), len(rel), np.median(rel), np.percentile(rel, 90), rel.max(), np.median(relb), relb.max()))
    d = np.array([float((got[True][k] - got[False][k]).norm() / got[False][k].norm()) for k in ref if float(ref[k].norm()) > 1e-9])
    print("seed %d  split precision vs fp32 launches: median %.2e  p90 %.2e  max %.2e" % (seed, np.median(d), np.percentile(d, 90), d.max()))
