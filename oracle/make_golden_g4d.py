"""G4d / G4e goldens (--shots 20 | 50): a 2000-step inner-loop trajectory at 20-shot (BASELINE configs[2] length: 5 epochs x 400 mini-batches of 5 over
the 2000 support view-images, finetune.py:270-299) run by the REFERENCE's backbone class + torch.optim.Adam on CPU: last-block
weight norms and probe features after 500 / 2000 steps in fp32, in fp64, and for other summation orders of the same fp32
arithmetic (1 ATen thread, oneDNN off) -- the reference's own spread, which is the envelope the HIP engine is held to
(round-4 verdict "missing 2": the 500-step envelope says nothing about 2000).  Build-container only; test infrastructure.

--shots 50 (G4e): the same at the 50-shot length, 5 x 1000 mini-batches = 5000 Adam steps (finetune_50.py:264-299), marks 2000 / 5000.

    python oracle/make_golden_g4d.py [--shots 20|50] [--threads 2]
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from meta_fine_tuning_amd import synthetic  # noqa: E402

CONFIGS = {20: (2000, (500, 2000), "g4d_inner_loop_2000.npz", 331, 78),        # n_shot: (steps, marks, file, episode seed, order seed)
           50: (5000, (2000, 5000), "g4e_inner_loop_5000.npz", 332, 79)}          # finetune_50.py:264-299 length (BASELINE configs[4])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--shots", type=int, default=20, choices=sorted(CONFIGS))
    args = ap.parse_args()
    N_STEPS, MARKS, fname, ep_seed, ord_seed = CONFIGS[args.shots]
    ns = args.shots
    mods = MG.import_reference()
    backbone = mods["backbone"]
    size = 84
    fac = MG.make_factory(backbone, size)
    views = synthetic.test_episode(ep_seed, 5, ns, 15, size, gen_examples=17)
    rs = np.random.RandomState(ord_seed)
    order = np.concatenate([rs.permutation(100 * ns) for _ in range(5)])        # 5 epochs over 5 * ns supports x (19 + 1) views
    out = {"order": order}
    variants = [("f32", torch.float32, args.threads, True), ("f64", torch.float64, args.threads, True),
                ("t1", torch.float32, 1, True), ("nodnn", torch.float32, args.threads, False)]
    for name, dt, nt, dnn in variants:
        torch.set_num_threads(nt)
        with torch.backends.mkldnn.flags(enabled=dnn):
            sd = synthetic.resnet10_state_dict(seed=9)
            m = fac()
            m.load_state_dict(sd)
            m = m.to(dt)
            names = [n for n, _ in m.named_parameters()]
            for n, p in m.named_parameters():
                if n in names[:-9]:
                    p.requires_grad = False
            opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=0.01)
            m.train()
            xa = torch.cat([v[:, :ns].contiguous().view(5 * ns, 3, size, size) for v in [views[0]] + views], 0).to(dt)
            ya = torch.from_numpy(np.tile(np.repeat(np.arange(5), ns), len(views) + 1))
            lossf = nn.CrossEntropyLoss()
            for step in range(N_STEPS):
                sel = torch.from_numpy(order[step * 5:(step + 1) * 5])
                opt.zero_grad()
                loss = lossf(m(xa[sel]), ya[sel])
                loss.backward()
                opt.step()
                if step + 1 in MARKS:
                    blk = m.trunk[7]
                    s = "_s%d_%s" % (step + 1, name)
                    out["wn_c1" + s] = np.array(float(blk.C1.weight.norm()))
                    out["wn_c2" + s] = np.array(float(blk.C2.weight.norm()))
                    out["wn_sc" + s] = np.array(float(blk.shortcut.weight.norm()))
                    out["loss" + s] = np.array(float(loss))
                    with torch.no_grad():
                        mm = copy.deepcopy(m)
                        mm.train()
                        out["probe" + s] = mm(xa[:5]).numpy()
        print("g4d/e", ns, name, {k: float(v) for k, v in out.items() if k.startswith("wn_") and k.endswith("_" + name)}, flush=True)
        np.savez(os.path.join(MG.GOLD, fname), variants=np.array([v[0] for v in variants]), **out)
    print("g4d done", flush=True)


if __name__ == "__main__":
    main()
