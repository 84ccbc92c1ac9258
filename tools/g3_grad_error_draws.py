"""Gradient error of one meta-training step (GnnNet.set_forward_loss(...).backward(), all 104 parameters) against the float64 oracle
over several (weights, episode) draws: relative L2 per tensor -> median / p90 / max.  A/B of a kernel form through the environment:
    gpurun -- bash -c 'python3 tools/g3_grad_error_draws.py; MFT_GEMM_RK_ROWS=0 python3 tools/g3_grad_error_draws.py'"""
import os, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import meta_fine_tuning_amd  # noqa: F401
from meta_fine_tuning_amd import synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet
from test_metatrain_gpu import _oracle_grads
torch.set_num_threads(8)
tag = "gemm_rk_rows=%s pair_rk_rows=%s" % (os.environ.get("MFT_GEMM_RK_ROWS", "default"), os.environ.get("MFT_PAIR_RK_ROWS", "default"))
allv = []
for draw, (ws, es) in enumerate(((7, 21), (8, 22), (9, 23), (10, 24))):
    sd = synthetic.gnnnet_state_dict(seed=ws)
    model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5); model.load_state_dict(sd); model = model.cuda(); model.train()
    x = synthetic.train_episode(es, 5, 5, 16, 84); model.n_query = 16
    loss = model.set_forward_loss(x); loss.backward()
    ref_loss, _, ref = _oracle_grads(sd, x)
    named = dict(model.named_parameters())
    rels = []
    for k, gr in ref.items():
        nrm = float(gr.norm())
        if nrm >= 1e-9:
            rels.append(float((named[k].grad.cpu().double() - gr).norm()) / nrm)
    v = np.array(rels); allv.append(v)
    print("%s draw %d: loss err %.2e; rel-L2 over %d tensors: median %.3e  p90 %.3e  max %.3e" % (tag, draw, abs(float(loss.detach()) - ref_loss), len(v), np.median(v), np.percentile(v, 90), v.max()), flush=True)
v = np.concatenate(allv)
print("%s ALL: median %.3e  p90 %.3e  max %.3e" % (tag, np.median(v), np.percentile(v, 90), v.max()), flush=True)
