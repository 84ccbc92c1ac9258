#!/usr/bin/env python3
"""Engine vs the reference's own FULL-LENGTH runs (goldens G19, oracle/make_golden_g19.py): 20-shot (2000 Adam steps per episode) and 50-shot
(5000 steps, finetune_50 + gnnnet_copy) per-episode accuracies, with the reference's oneDNN-off re-run of the first episodes as the yardstick.
    gpurun -- python tools/accuracy_g19.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_accuracy_gpu as T

gd = os.path.join(ROOT, "tests", "golden")
for name, fname, ns, batch, fold in (("20-shot, 2000 steps/episode, E = 96", "g19_accuracy_20shot.npz", 20, 96, False),
                                     ("50-shot, 5000 steps/episode, E = 64", "g19_accuracy_50shot.npz", 50, 64, True)):
    accs, ref, spread = T._run_g19(gd, fname, ns, batch, fold)
    d = np.abs(accs - ref)
    k = len(spread)
    ds = np.abs(spread - ref[:k])
    print("%s: %d episodes" % (name, len(ref)))
    print("  mean accuracy: engine %.3f %%  reference %.3f %%  (difference %.3f; bar %.3f)" % (
        accs.mean(), ref.mean(), abs(accs.mean() - ref.mean()), 0.2 * (600.0 / len(ref)) ** 0.5 + 0.1))
    print("  per episode |engine - reference|: mean %.2f, median %.2f, 90th percentile %.2f, max %.2f points; identical: %d of %d" % (
        d.mean(), np.median(d), np.percentile(d, 90), d.max(), int((d < 1e-9).sum()), len(d)))
    print("  the reference against ITSELF with oneDNN off (first %d episodes): mean %.2f, max %.2f points; identical: %d of %d" % (
        k, ds.mean(), ds.max(), int((ds < 1e-9).sum()), k))
