"""G9 golden vectors: per-episode accuracies of the REFERENCE's own finetune() on structured synthetic episodes.

Build-container only (imports /root/reference on CPU, recipe of oracle/make_golden.py).  Two steps:
  1. meta-train the few-shot head (fc + gnn; the backbone stays at its seeded random init) with the reference's own
     GnnNet.set_forward_loss on synthetic meta-train episodes, so that accuracies are meaningful (a random head
     predicts a fixed wrong permutation of the labels).  The trained head (1.6 MB) is a fixture: tests/golden/g9_head.npz.
  2. run the reference finetune() over the episode list and store per-episode accuracies + a score checksum:
     config A: fine_tune_epoch=1, gen_examples=2, 600 episodes; config B: fine_tune_epoch=5, gen_examples=17 (BASELINE configs[1]), 40 episodes.
Episodes and backbone weights are regenerated from seeds by meta-fine-tuning_amd/synthetic.py at test time.
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from meta_fine_tuning_amd import synthetic  # noqa: E402

NOISE = 2.0            # per-pixel noise of the synthetic classes (default 1.0 is trivially separable; 2.0 gives ~79 % nearest-prototype accuracy)
SEED_SD = 31
EP_SEED0 = 90000


def g9_episode(i, gen_examples):
    return synthetic.test_episode(EP_SEED0 + i, 5, 5, 15, 84, gen_examples=gen_examples, noise=NOISE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--train-episodes", type=int, default=300)
    ap.add_argument("--nA", type=int, default=600)
    ap.add_argument("--nB", type=int, default=40)
    ap.add_argument("--only", default="AB", help="which configs to (re)compute; the other one is kept from the existing file")
    ap.add_argument("--resume", action="store_true",
                    help="continue a config from the episodes already in g9_accuracy.npz: the global numpy stream is fast-forwarded "
                         "by the permutations those episodes drew (finetune.py:272: fine_tune_epoch draws of permutation(n_way*n_support*(G+3)) "
                         "each) and the last stored episode is recomputed as a check of the stream position")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    mods = MG.import_reference()
    backbone, gnnnet, finetune = mods["backbone"], mods["methods.gnnnet"], mods["finetune"]
    size = 84
    fac = MG.make_factory(backbone, size)
    sd = synthetic.gnnnet_state_dict(seed=SEED_SD)

    head_path = os.path.join(MG.GOLD, "g9_head.npz")
    if not os.path.exists(head_path):
        torch.manual_seed(0)
        model = gnnnet.GnnNet(fac, n_way=5, n_support=5)
        model.load_state_dict(sd)
        model.train()
        model.n_query = 16
        head = list(model.fc.parameters()) + list(model.gnn.parameters())
        opt = torch.optim.Adam(head, lr=1e-3)
        t0 = time.time()
        for it in range(args.train_episodes):
            x = synthetic.train_episode(70000 + it, 5, 5, 16, size, noise=NOISE)
            opt.zero_grad()
            loss = model.set_forward_loss(x)
            loss.backward()
            opt.step()
            if it % 25 == 0:
                print("head train %d loss %.4f (%.0fs)" % (it, float(loss), time.time() - t0), flush=True)
        out = {k: v.detach().numpy() for k, v in model.state_dict().items() if k.startswith(("fc.", "gnn."))}
        np.savez_compressed(head_path, **out)
    hz = np.load(head_path)
    for k in hz.files:
        sd[k] = torch.from_numpy(hz[k])

    finetune.model_dict["ResNet10"] = fac
    y = np.repeat(np.arange(5), 15)
    res = {}
    acc_path = os.path.join(MG.GOLD, "g9_accuracy.npz")
    if os.path.exists(acc_path):
        old = np.load(acc_path)
        res = {k: old[k] for k in old.files if k[:4] in ("acc_", "chk_", "cfg_")}
    for tag, E, G, n in (("A", 1, 2, args.nA), ("B", 5, 17, args.nB)):
        if tag not in args.only:
            continue
        finetune.params = argparse.Namespace(model="ResNet10", fine_tune_epoch=E)
        accs, chk = [], []
        np.random.seed(10)                     # finetune.py:425
        t0 = time.time()
        start = 0
        if args.resume and ("acc_" + tag) in res and len(res["acc_" + tag]) > 1:
            accs, chk = list(res["acc_" + tag]), list(res["chk_" + tag])
            start = len(accs) - 1                # recompute the last stored episode: must reproduce its checksum
            for _ in range(start * E):
                np.random.permutation(25 * (G + 3))
            last_chk = chk[-1]
            accs, chk = accs[:-1], chk[:-1]
        for i in range(start, n):
            model = gnnnet.GnnNet(fac, n_way=5, n_support=5)
            model.load_state_dict(sd)
            model.train()
            liz = g9_episode(i, G)
            sc = finetune.finetune(liz, None, model, copy.deepcopy(sd), None, n_query=15, n_way=5, n_support=5)
            sc = sc.detach().numpy()
            accs.append(float((sc.argmax(1) == y).mean() * 100.0))
            chk.append(sc[:, 0].astype(np.float64).sum())
            if args.resume and i == start and start > 0:
                assert abs(chk[-1] - last_chk) < 1e-4, ("resume: stream position check failed", chk[-1], last_chk)
                print("resume check ok at episode %d (chk %.9f)" % (i, chk[-1]), flush=True)
            if i % 20 == 0 or i == n - 1:
                print("config %s episode %d acc %.2f mean %.2f (%.0fs)" % (tag, i, accs[-1], np.mean(accs), time.time() - t0), flush=True)
                part = dict(res)
                part["acc_" + tag], part["chk_" + tag], part["cfg_" + tag] = np.array(accs), np.array(chk), np.array([E, G, len(accs)])
                np.savez(acc_path, noise=np.array(NOISE), seed_sd=np.array(SEED_SD), ep_seed0=np.array(EP_SEED0), **part)
        res["acc_" + tag] = np.array(accs)
        res["chk_" + tag] = np.array(chk)
        res["cfg_" + tag] = np.array([E, G, n])
        np.savez(os.path.join(MG.GOLD, "g9_accuracy.npz"), noise=np.array(NOISE), seed_sd=np.array(SEED_SD),
                 ep_seed0=np.array(EP_SEED0), **res)
        print("config %s: mean acc %.3f +- %.3f" % (tag, np.mean(accs), 1.96 * np.std(accs) / np.sqrt(n)), flush=True)


if __name__ == "__main__":
    main()
