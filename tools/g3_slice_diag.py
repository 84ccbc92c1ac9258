import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import meta_fine_tuning_amd
from meta_fine_tuning_amd import synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet
from test_metatrain_gpu import _oracle_grads
torch.set_num_threads(8)
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/g3_gnnnet_set_forward.npz"))
sd = synthetic.gnnnet_state_dict(seed=7)
model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5); model.load_state_dict(sd); model = model.cuda(); model.train()
x = synthetic.train_episode(21, 5, 5, 16, 84); model.n_query = 16
loss = model.set_forward_loss(x); loss.backward()
_, _, ref = _oracle_grads(sd, x)
named = dict(model.named_parameters())
for key, sl, gk in (("fc.0.weight", (slice(0, 4), slice(0, 8)), "grad_fc0w_slice"), ("feature.trunk.7.C2.weight", (slice(0, 2), slice(0, 4), 1, 1), "grad_c7c2_slice"),
                    ("feature.trunk.0.weight", (slice(0, 2), slice(None), 3, 3), "grad_stem_slice")):
    ours = named[key].grad[sl].cpu().double().numpy(); o64 = ref[key][sl].numpy(); gold = g[gk].astype(np.float64)
    print(os.environ.get("MFT_GEMM_RK_ROWS", "default"), key, "ours-o64 %.3e  gold-o64 %.3e  ours-gold %.3e" % (np.abs(ours - o64).max(), np.abs(gold - o64).max(), np.abs(ours - gold).max()), flush=True)
rels = {}
for k, gr in ref.items():
    nrm = float(gr.norm())
    if nrm < 1e-9:
        continue
    rels[k] = float((named[k].grad.cpu().double() - gr).norm()) / nrm
v = np.array(sorted(rels.values()))
print(os.environ.get("MFT_GEMM_RK_ROWS", "default"), "rel-L2 over %d tensors: median %.3e  p90 %.3e  max %.3e (%s)" % (len(v), np.median(v), np.percentile(v, 90), v.max(), max(rels, key=rels.get)), flush=True)
print(os.environ.get("MFT_GEMM_RK_ROWS", "default"), "loss", float(loss.detach()), "gold", float(g["loss"]), flush=True)
