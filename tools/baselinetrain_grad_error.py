#!/usr/bin/env python3
"""BaselineTrain backbone-gradient error of the HIP path vs float64, next to torch's own fp32 CPU kernels vs float64 on the SAME
draw (bound the HIP path by a multiple of the fp32 reference's own distance instead of a flat loose bound).
Prints, per seed, the worst / median ratio over the backbone's tensors.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
import __graft_entry__ as ge

ge.build()
from meta_fine_tuning_amd import backbone, synthetic
from meta_fine_tuning_amd.methods.baselinetrain import BaselineTrain
from oracle import mft_oracle as O

torch.set_num_threads(8)
for seed in (21, 22, 23, 24, 25):
    torch.manual_seed(3)
    m = BaselineTrain(backbone.ResNet10, num_class=200).cuda()
    sd = synthetic.resnet10_state_dict(seed=53)
    m.feature.load_state_dict(sd)
    m.train()
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((16, 3, 84, 84)).astype(np.float32))
    y = torch.from_numpy(rs.randint(0, 200, size=16))
    m.forward_loss(x, y).backward()
    refs = {}
    for dt in (torch.float64, torch.float32):
        fsd = {k: v.to(dt).requires_grad_(v.is_floating_point() and "running" not in k) for k, v in O.clone_state(sd).items()}
        w = m.classifier.weight.detach().cpu().to(dt).requires_grad_(True)
        b = m.classifier.bias.detach().cpu().to(dt).requires_grad_(True)
        feat = O.resnet10_forward(fsd, x.to(dt), "", train=True, track=False)
        F.cross_entropy(F.linear(feat, w, b), y).backward()
        refs[dt] = {k: v.grad.double() for k, v in fsd.items() if v.grad is not None}
    # the same model on torch's OWN GPU kernels (MIOpen / rocBLAS, fp32): another fp32 GPU implementation of the same arithmetic
    fsd = {k: v.cuda().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in O.clone_state(sd).items()}
    w = m.classifier.weight.detach().clone().requires_grad_(True)
    b = m.classifier.bias.detach().clone().requires_grad_(True)
    feat = O.resnet10_forward(fsd, x.cuda(), "", train=True, track=False)
    F.cross_entropy(F.linear(feat, w, b), y.cuda()).backward()
    refs["gpu"] = {k: v.grad.double().cpu() for k, v in fsd.items() if v.grad is not None}
    rows = []
    eg = []
    for name, p in m.feature.named_parameters():
        g64 = refs[torch.float64][name]
        if float(g64.norm()) < 1e-9:
            continue
        e_hip = float((p.grad.cpu().double() - g64).norm() / g64.norm())
        e_32 = float((refs[torch.float32][name] - g64).norm() / g64.norm())
        rows.append((name, e_hip, e_32, e_hip / max(e_32, 1e-12)))
        eg.append(float((refs["gpu"][name] - g64).norm() / g64.norm()))
    worst = max(rows, key=lambda r: r[3])
    print("seed %d: worst ratio %.2f (%s: hip %.2e, torch-fp32 %.2e); median ratio %.2f; max hip %.2e; max torch32 %.2e; stem hip %.2e torch32 %.2e"
          % (seed, worst[3], worst[0], worst[1], worst[2], float(np.median([r[3] for r in rows])), max(r[1] for r in rows), max(r[2] for r in rows),
             rows[0][1], rows[0][2]), flush=True)
    print("        torch-GPU-fp32 vs float64: max %.2e median %.2e stem %.2e;  hip median %.2e" % (max(eg), float(np.median(eg)), eg[0],
          float(np.median([r[1] for r in rows]))), flush=True)
