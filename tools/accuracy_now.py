#!/usr/bin/env python3
"""Engine accuracies against the reference's per-episode accuracies (golden G9) with the current kernels.
Usage: accuracy_now.py [A|B|AB] [batch]   (MFT_CONV_KNOBS=9003 selects the exact division / square-root Adam epilogue)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine as eng, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "AB"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for knob in os.environ.get("MFT_CONV_KNOBS", "").split(","):
    if knob.strip():
        _lib.lib().mft_debug_set_conv_tile(int(knob))
gd = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gd, "g9_accuracy.npz"))
sd = synthetic.gnnnet_state_dict(seed=int(g["seed_sd"]))
hz = np.load(os.path.join(gd, "g9_head.npz"))
for k in hz.files:
    sd[k] = torch.from_numpy(hz[k])
y = np.repeat(np.arange(5), 15)
for tag in which:
    E_ep, G, _ = [int(v) for v in g["cfg_" + tag]]
    ref = g["acc_" + tag]
    n = len(ref)
    e = eng.FinetuneEngine(sd, 5, 5, 15, 84, n_views=2 + G, fine_tune_epoch=E_ep, episodes_per_batch=batch)
    accs = []
    np.random.seed(10)
    with ThreadPoolExecutor(max_workers=12) as ex:
        for i in range(0, n, batch):
            eps = list(ex.map(lambda j: synthetic.test_episode(int(g["ep_seed0"]) + j, 5, 5, 15, 84, gen_examples=G, noise=float(g["noise"])),
                              range(i, min(i + batch, n))))
            sc = e.run_batch(eps).cpu().numpy()
            accs += [float((s.argmax(1) == y).mean() * 100.0) for s in sc]
    accs = np.array(accs)
    d = np.abs(accs - ref)
    worst = np.argsort(-d)[:5]
    print("config %s: %d episodes: engine mean %.3f  reference mean %.3f  identical %.3f  p90 %.2f  p99 %.2f  max %.2f  worst %s" % (
        tag, n, accs.mean(), ref.mean(), np.mean(d < 1e-6), np.percentile(d, 90), np.percentile(d, 99), d.max(),
        [(int(i), float(accs[i]), float(ref[i])) for i in worst]))
    e.close()
