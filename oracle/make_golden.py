"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Build-container only (needs /root/reference; never runs on the GPU box).  It
imports the reference's own modules (recipe: SURVEY.md §8(c)), loads the
in-repo deterministic weights/episodes into them, runs the hot-path calls and
stores *outputs only* (inputs/weights are regenerated from seeds by
``meta-fine-tuning_amd/synthetic.py``).  No reference source is copied.

    python oracle/make_golden.py            # writes tests/golden/*.npz
"""
import argparse
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
import meta_fine_tuning_amd  # noqa: E402  (import shim at repo root)
from meta_fine_tuning_amd import synthetic  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    """SURVEY.md §8(c) recipe: .cuda() no-ops + stub torchvision/h5py."""
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda s, *a, **k: s
    nn.Module.cuda = lambda s, *a, **k: s
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.datasets = sys.modules["torchvision.datasets"]
    tv.transforms.ToTensor = object
    for a in ("ImageFolder", "CIFAR10", "CIFAR100"):
        setattr(tv.datasets, a, object)
    import backbone  # noqa
    import methods.gnn  # noqa
    import methods.gnnnet  # noqa
    import methods.gnnnet_copy  # noqa
    import methods.baselinefinetune  # noqa
    import finetune  # noqa
    return sys.modules


def ref_resnet10(backbone, size):
    m = backbone.ResNet10(flatten=True)
    if size != 224:
        # the unmodified reference cannot run 84x84 (AvgPool2d(7) on 3x3).
        m.trunk[8] = nn.AvgPool2d(size // 32 + (1 if size % 32 else 0))
    return m


def pool_for(size):
    h = size
    h = (h + 6 - 7) // 2 + 1
    h = (h + 2 - 3) // 2 + 1
    for _ in range(3):
        h = (h + 2 - 3) // 2 + 1
    return h


def make_factory(backbone, size):
    def f(flatten=True):
        m = backbone.ResNet10(flatten)
        if size != 224:
            m.trunk[8] = nn.AvgPool2d(pool_for(size))
        return m
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    mods = import_reference()
    backbone = mods["backbone"]
    gnn = mods["methods.gnn"]
    gnnnet = mods["methods.gnnnet"]
    gnnnet_copy = mods["methods.gnnnet_copy"]
    blf = mods["methods.baselinefinetune"]
    finetune = mods["finetune"]
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)

    def want(name):
        return (not args.only) or (name in args.only.split(","))

    # ---------------------------------------------------------------- G1 ResNet10 forward
    if want("g1"):
        out = {}
        for size in (84, 224):
            sd = synthetic.resnet10_state_dict(seed=3)
            m = make_factory(backbone, size)()
            m.load_state_dict(sd)
            m.train()
            x = synthetic.train_episode(11, 5, 1, 0, size).view(5, 3, size, size)
            taps = {}
            for name, mod in m.named_modules():
                if isinstance(mod, nn.Conv2d) or name in ("trunk.3", "trunk.4", "trunk.5", "trunk.6", "trunk.7"):
                    mod.register_forward_hook(
                        lambda mod_, i_, o_, n=name: taps.__setitem__(n, (float(o_.mean()), float(o_.norm()))))
            with torch.no_grad():
                f = m(x)
            out["feat_%d" % size] = f.numpy()
            st = m.state_dict()
            out["rm1_%d" % size] = st["trunk.1.running_mean"].numpy()
            out["rv1_%d" % size] = st["trunk.1.running_var"].numpy()
            out["rm7_%d" % size] = st["trunk.7.BN2.running_mean"].numpy()
            out["rv7_%d" % size] = st["trunk.7.BN2.running_var"].numpy()
            out["nbt_%d" % size] = st["trunk.7.BN2.num_batches_tracked"].numpy()
            names = sorted(taps)
            out["tapnames_%d" % size] = np.array(names)
            out["taps_%d" % size] = np.array([taps[n] for n in names], dtype=np.float64)
        np.savez(os.path.join(GOLD, "g1_resnet10_fwd.npz"), **out)
        print("g1 done")

    # ---------------------------------------------------------------- G2 GNN_nl
    if want("g2"):
        out = {}
        sd = synthetic.gnn_head_state_dict(seed=5)
        gsd = {k[len("gnn."):]: v for k, v in sd.items() if k.startswith("gnn.")}
        g = gnn.GNN_nl(133, 96, 5)
        g.load_state_dict(gsd)
        g.train()
        for (B, N) in ((15, 30), (16, 30), (2, 105), (2, 130)):
            rs = np.random.RandomState(100 + N + B)
            nodes = torch.from_numpy(rs.standard_normal((B, N, 133)).astype(np.float32))
            with torch.no_grad():
                o = g(nodes)
                W0 = g.layer_w0(nodes, torch.eye(N).unsqueeze(0).repeat(B, 1, 1).unsqueeze(3))
            out["out_%d_%d" % (B, N)] = o.numpy()
            out["A0_%d_%d" % (B, N)] = W0[..., 1].numpy()
        np.savez(os.path.join(GOLD, "g2_gnn.npz"), **out)
        print("g2 done")

    # ---------------------------------------------------------------- G3 GnnNet.set_forward (+grads)
    if want("g3"):
        out = {}
        size = 84
        sd = synthetic.gnnnet_state_dict(seed=7)
        model = gnnnet.GnnNet(make_factory(backbone, size), n_way=5, n_support=5)
        model.load_state_dict(sd)
        model.train()
        x = synthetic.train_episode(21, 5, 5, 16, size)
        model.n_query = 16
        loss = model.set_forward_loss(x)
        loss.backward()
        with torch.no_grad():
            model2 = gnnnet.GnnNet(make_factory(backbone, size), n_way=5, n_support=5)
            model2.load_state_dict(sd)
            model2.train()
            model2.n_query = 16
            scores = model2.set_forward(x)
        out["scores"] = scores.numpy()
        out["loss"] = np.array(float(loss))
        gn = {}
        for n, p in model.named_parameters():
            gn[n] = float(p.grad.norm())
        names = sorted(gn)
        out["gradnames"] = np.array(names)
        out["gradnorms"] = np.array([gn[n] for n in names])
        out["grad_fc0w_slice"] = model.fc[0].weight.grad[:4, :8].numpy()
        out["grad_last_slice"] = model.gnn.layer_last.fc.weight.grad[:, :8].numpy()
        out["grad_c7c2_slice"] = model.feature.trunk[7].C2.weight.grad[:2, :4, 1, 1].numpy()
        out["grad_stem_slice"] = model.feature.trunk[0].weight.grad[:2, :, 3, 3].numpy()
        np.savez(os.path.join(GOLD, "g3_gnnnet_set_forward.npz"), **out)
        print("g3 done")

    # ---------------------------------------------------------------- G4 inner-loop trajectory
    if want("g4"):
        out = {}
        size = 84
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            sd = synthetic.resnet10_state_dict(seed=9)
            m = make_factory(backbone, size)()
            m.load_state_dict(sd)
            m = m.to(dt)
            names = [n for n, _ in m.named_parameters()]
            for n, p in m.named_parameters():
                if n in names[:-9]:
                    p.requires_grad = False
            opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=0.01)
            m.train()
            views = synthetic.test_episode(31, 5, 5, 15, size, gen_examples=1)
            xa = torch.cat([v[:, :5].contiguous().view(25, 3, size, size) for v in [views[0]] + views], 0).to(dt)
            ya = torch.from_numpy(np.tile(np.repeat(np.arange(5), 5), len(views) + 1))
            rs = np.random.RandomState(77)
            perm = rs.permutation(xa.shape[0])
            lossf = nn.CrossEntropyLoss()
            probe = xa[:5]
            for step in range(8):
                sel = torch.from_numpy(perm[step * 5:(step + 1) * 5])
                opt.zero_grad()
                o = m(xa[sel])
                loss = lossf(o, ya[sel])
                loss.backward()
                if step == 0:
                    out["feat0_" + tag] = o.detach().numpy()
                    out["loss0_" + tag] = np.array(float(loss))
                    blk = m.trunk[7]
                    out["g_c1_slice_" + tag] = blk.C1.weight.grad[:2, :4].numpy().copy()
                    out["g_c2_slice_" + tag] = blk.C2.weight.grad[:2, :4].numpy().copy()
                    out["g_sc_slice_" + tag] = blk.shortcut.weight.grad[:4, :8, 0, 0].numpy().copy()
                    for nm, mod in (("bn1", blk.BN1), ("bn2", blk.BN2), ("bnsc", blk.BNshortcut)):
                        out["g_%s_w_%s" % (nm, tag)] = mod.weight.grad.numpy().copy()
                        out["g_%s_b_%s" % (nm, tag)] = mod.bias.grad.numpy().copy()
                    out["gn_c1_" + tag] = np.array(float(blk.C1.weight.grad.norm()))
                    out["gn_c2_" + tag] = np.array(float(blk.C2.weight.grad.norm()))
                    out["gn_sc_" + tag] = np.array(float(blk.shortcut.weight.grad.norm()))
                opt.step()
                if step in (0, 6):
                    blk = m.trunk[7]
                    out["w_c2_slice_s%d_%s" % (step + 1, tag)] = blk.C2.weight.detach()[:2, :4].numpy().copy()
                    out["wn_c1_s%d_%s" % (step + 1, tag)] = np.array(float(blk.C1.weight.norm()))
                    out["wn_c2_s%d_%s" % (step + 1, tag)] = np.array(float(blk.C2.weight.norm()))
                    out["bn2_w_s%d_%s" % (step + 1, tag)] = blk.BN2.weight.detach().numpy().copy()
                    with torch.no_grad():
                        mm = copy.deepcopy(m)
                        mm.train()
                        out["probe_s%d_%s" % (step + 1, tag)] = mm(probe).numpy()
        out["perm"] = perm
        np.savez(os.path.join(GOLD, "g4_inner_loop.npz"), **out)
        print("g4 done")

    # ---------------------------------------------------------------- G5 finetune() scores
    if want("g5"):
        out = {}
        size = 84
        sd = synthetic.gnnnet_state_dict(seed=13)
        finetune.model_dict["ResNet10"] = make_factory(backbone, size)
        for (E, G) in ((0, 0), (1, 0), (1, 2), (2, 1)):
            finetune.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=E)
            model = gnnnet.GnnNet(make_factory(backbone, size), n_way=5, n_support=5)
            model.load_state_dict(sd)
            model.train()
            liz = synthetic.test_episode(41 + G, 5, 5, 15, size, gen_examples=G)
            np.random.seed(10)
            sc = finetune.finetune(liz, None, model, copy.deepcopy(sd), None, n_query=15, n_way=5, n_support=5)
            out["scores_E%d_G%d" % (E, G)] = sc.numpy()
        np.savez(os.path.join(GOLD, "g5_finetune.npz"), **out)
        print("g5 done")

    # ---------------------------------------------------------------- G6 first-order MAML algebra
    if want("g6"):
        out = {}
        size = 84
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            sd = synthetic.gnnnet_state_dict(seed=17)
            model = gnnnet.GnnNet(make_factory(backbone, size), n_way=5, n_support=5)
            model.load_state_dict(sd)
            model = model.to(dt)
            model.support_label = model.support_label.to(dt)
            model.train()
            opt = torch.optim.Adam(model.parameters())
            np.random.seed(10)
            for it in range(2):
                x = synthetic.train_episode(51 + it, 5, 5, 16, size).to(dt)
                model.n_query = 16
                opt.zero_grad()
                loss = model.set_forward_loss_finetune(x)
                loss.backward()
                opt.step()
                out["loss_%d_%s" % (it, tag)] = np.array(float(loss.detach()))
                out["c2_slice_%d_%s" % (it, tag)] = model.feature.trunk[7].C2.weight.detach()[:2, :4, 1, 1].numpy().copy()
                out["c2n_%d_%s" % (it, tag)] = np.array(float(model.feature.trunk[7].C2.weight.norm()))
                out["stemn_%d_%s" % (it, tag)] = np.array(float(model.feature.trunk[0].weight.norm()))
                out["f3_c2n_%d_%s" % (it, tag)] = np.array(float(model.feature3.trunk[7].C2.weight.norm()))
                out["f2_c2n_%d_%s" % (it, tag)] = np.array(float(model.feature2.trunk[7].C2.weight.norm()))
                out["fc0_slice_%d_%s" % (it, tag)] = model.fc[0].weight.detach()[:2, :8].numpy().copy()
            model.MAML_update()
            out["c2_slice_final_" + tag] = model.feature.trunk[7].C2.weight.detach()[:2, :4, 1, 1].numpy().copy()
        np.savez(os.path.join(GOLD, "g6_maml.npz"), **out)
        print("g6 done")

    # ---------------------------------------------------------------- G7 gnnnet_copy (50-shot fold)
    if want("g7"):
        out = {}
        sd = synthetic.gnn_head_state_dict(seed=19)
        model = gnnnet_copy.GnnNet(make_factory(backbone, 84), n_way=5, n_support=50)
        st = model.state_dict()
        st.update(sd)
        model.load_state_dict(st)
        model.train()
        model.n_query = 15
        rs = np.random.RandomState(61)
        feats = torch.from_numpy(rs.standard_normal((5, 65, 512)).astype(np.float32))
        with torch.no_grad():
            sc = model.set_forward(feats, is_feature=True)
        out["scores"] = sc.numpy()
        np.savez(os.path.join(GOLD, "g7_gnnnet50.npz"), **out)
        print("g7 done")

    # ---------------------------------------------------------------- G8 BaselineFinetune
    if want("g8"):
        out = {}
        model = blf.BaselineFinetune(make_factory(backbone, 84), n_way=5, n_support=5)
        model.n_query = 15
        rs = np.random.RandomState(71)
        feats = torch.from_numpy(rs.standard_normal((5, 20, 512)).astype(np.float32))
        torch.manual_seed(123)
        lin = nn.Linear(512, 5)
        out["w0"] = lin.weight.detach().numpy().copy()
        out["b0"] = lin.bias.detach().numpy().copy()
        torch.manual_seed(123)
        np.random.seed(10)
        sc = model.set_forward(feats, is_feature=True)
        out["scores"] = sc.detach().numpy()
        np.savez(os.path.join(GOLD, "g8_baselinefinetune.npz"), **out)
        print("g8 done")

    # ---------------------------------------------------------------- G10 finetune_linear + "all" ensemble
    if want("g10"):
        out = {}
        size = 84
        sd = synthetic.gnnnet_state_dict(seed=37)
        finetune.model_dict["ResNet10"] = make_factory(backbone, size)
        finetune.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=1)
        rs = np.random.RandomState(81)
        w0 = (rs.uniform(-1, 1, size=(5, 512)) / np.sqrt(512)).astype(np.float32)
        b0 = (rs.uniform(-1, 1, size=(5,)) / np.sqrt(512)).astype(np.float32)
        RefLinear = nn.Linear

        class SeededLinear(RefLinear):                   # finetune.Classifier's nn.Linear(512, 5) with known initial weights
            def __init__(self, i, o, *a, **k):           # instead of torch's RNG draw (finetune.py:36-38,65)
                super().__init__(i, o, *a, **k)
                if (o, i) == w0.shape:
                    with torch.no_grad():
                        self.weight.copy_(torch.from_numpy(w0))
                        self.bias.copy_(torch.from_numpy(b0))

        nn.Linear = SeededLinear
        liz = synthetic.test_episode(91, 5, 5, 15, size, gen_examples=1)
        np.random.seed(10)
        sc_lin = finetune.finetune_linear(liz, None, state_in=copy.deepcopy(sd), linear=True, save_it=None, n_query=15,
                                          n_way=5, n_support=5)
        out["w0"], out["b0"] = w0, b0
        out["scores_linear"] = sc_lin.numpy()
        nn.Linear = RefLinear
        # finetune.py:647-649 ("--method all"): scores_out = finetune_linear(...); scores_out += finetune(...), same numpy stream
        model = gnnnet.GnnNet(make_factory(backbone, size), n_way=5, n_support=5)
        model.load_state_dict(sd)
        model.train()
        sc_gnn = finetune.finetune(liz, None, model, copy.deepcopy(sd), None, n_query=15, n_way=5, n_support=5)
        out["scores_all"] = (sc_lin + sc_gnn).numpy()
        np.savez(os.path.join(GOLD, "g10_finetune_linear.npz"), **out)
        print("g10 done")

    # ---------------------------------------------------------------- G11 finetune(freeze_backbone=True)
    if want("g11"):
        out = {}
        size = 84
        sd = synthetic.gnnnet_state_dict_with_running_stats(seed=47)
        finetune.model_dict["ResNet10"] = make_factory(backbone, size)
        finetune.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=2)
        model = gnnnet.GnnNet(make_factory(backbone, size), n_way=5, n_support=5)
        model.load_state_dict(sd)
        model.train()
        liz = synthetic.test_episode(95, 5, 5, 15, size, gen_examples=1)
        np.random.seed(10)
        sc = finetune.finetune(liz, None, model, copy.deepcopy(sd), None, n_query=15, freeze_backbone=True, n_way=5, n_support=5)
        out["scores"] = sc.numpy()
        out["next_perm"] = np.random.permutation(7)          # position of the numpy stream after the call
        np.savez(os.path.join(GOLD, "g11_finetune_frozen.npz"), **out)
        print("g11 done")

    # ---------------------------------------------------------------- G12 finetune_linear(freeze_backbone=True)
    if want("g12"):
        out = {}
        size = 84
        sd = synthetic.gnnnet_state_dict_with_running_stats(seed=57)
        finetune.model_dict["ResNet10"] = make_factory(backbone, size)
        finetune.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=1)
        rs = np.random.RandomState(83)
        w0 = (rs.uniform(-1, 1, size=(5, 512)) / np.sqrt(512)).astype(np.float32)
        b0 = (rs.uniform(-1, 1, size=(5,)) / np.sqrt(512)).astype(np.float32)
        RefLinear = nn.Linear

        class SeededLinear12(RefLinear):
            def __init__(self, i, o, *a, **k):
                super().__init__(i, o, *a, **k)
                if (o, i) == w0.shape:
                    with torch.no_grad():
                        self.weight.copy_(torch.from_numpy(w0))
                        self.bias.copy_(torch.from_numpy(b0))

        nn.Linear = SeededLinear12
        liz = synthetic.test_episode(97, 5, 5, 15, size, gen_examples=1)
        np.random.seed(10)
        sc = finetune.finetune_linear(liz, None, state_in=copy.deepcopy(sd), linear=True, save_it=None, n_query=15,
                                      freeze_backbone=True, n_way=5, n_support=5)
        nn.Linear = RefLinear
        out["w0"], out["b0"] = w0, b0
        out["scores"] = sc.numpy()
        out["next_perm"] = np.random.permutation(7)          # position of the numpy stream after the call
        np.savez(os.path.join(GOLD, "g12_finetune_linear_frozen.npz"), **out)
        print("g12 done")


if __name__ == "__main__":
    main()
