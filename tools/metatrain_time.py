#!/usr/bin/env python3
"""One meta-training step (set_forward_loss + backward + fused outer Adam; meta_template.py:76-92) at the 5-shot graph size
(N = 30, 16 graphs: BASELINE configs[3]) and at the 50-shot compressed graph (gnnnet_copy: N = 130, 16 graphs: train_50.py), with
the upper-triangle fused Wcompute backward (default) and with round 2's materialised form (tools/legacy_wcompute_bwd.py).
    python tools/metatrain_time.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import optim, synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods import gnnnet_copy
from meta_fine_tuning_amd.methods.gnnnet import GnnNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def run(tag, n_shot, cls):
    torch.manual_seed(0)
    model = cls(model_dict["ResNet10"], n_way=5, n_support=n_shot).cuda()
    model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
    model.train()
    model.n_query = 16
    opt = optim.Adam(model.parameters())
    eps = [synthetic.train_episode(5000 + i, 5, n_shot, 16, 84).cuda() for i in range(2)]

    def step(i):
        opt.zero_grad()
        loss = model.set_forward_loss(eps[i % 2])
        loss.backward()
        opt.step()
        return loss
    for i in range(3):
        loss = step(i)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-34s %d-shot  %8.2f ms per step   peak memory %7.1f MB   loss %.5f" % (tag, n_shot, dt * 1e3, torch.cuda.max_memory_allocated() / 1e6,
                                                                                float(loss)))


for legacy in (False, True, False):
    if legacy:
        import legacy_wcompute_bwd
        import importlib
        legacy_wcompute_bwd.install()
    else:
        from meta_fine_tuning_amd import functional_bwd as FB
        import importlib
        importlib.reload(FB)
        from meta_fine_tuning_amd import autograd_ops
        autograd_ops.FB = FB
    tag = "materialised N*N rows (round 2)" if legacy else "fused upper-triangle rows"
    run(tag, 5, GnnNet)
    run(tag, 50, gnnnet_copy.GnnNet)
