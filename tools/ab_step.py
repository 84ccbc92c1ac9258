#!/usr/bin/env python3
"""Same-lease A/B of the lockstep inner step under two sets of debug knobs, alternated in ONE process (box-to-box spread is larger
than most effects), with socket power and shader clock sampled beside the timing (the step is power-limited).
Usage: ab_step.py "A knobs" "B knobs" [alternations] [E] [steps]
  knobs: comma-separated codes, cNNNN = mft_debug_set_conv_tile(NNNN), xNNN = mft_debug_set_x3_tile(NNN); "" = defaults
  e.g.   ab_step.py "" "c9600"        (default against the padded weight-gradient MFMA loop)"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine as eng, synthetic

A = sys.argv[1] if len(sys.argv) > 1 else ""
B = sys.argv[2] if len(sys.argv) > 2 else ""
ALT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
E = int(sys.argv[4]) if len(sys.argv) > 4 else 128
STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 300
dev = "cuda:0"
lib = _lib.lib()

CARDS = []
for card in sorted(glob.glob("/sys/class/drm/card*/device")):
    hw = glob.glob(card + "/hwmon/hwmon*")
    if hw and os.path.exists(hw[0] + "/power1_input"):
        CARDS.append((hw[0] + "/power1_input", hw[0] + "/freq1_input"))
samples, stop = [], False


def rd(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return float("nan")


def sampler():
    while not stop:
        samples.append((time.time(), [(rd(pw) * 1e-6, rd(fq) * 1e-6) for pw, fq in CARDS]))
        time.sleep(0.02)


def apply(knobs):
    lib.mft_debug_reset()
    for k in knobs.split(","):
        k = k.strip()
        if k.startswith("c"):
            lib.mft_debug_set_conv_tile(int(k[1:]))
        elif k.startswith("x"):
            lib.mft_debug_set_x3_tile(int(k[1:]))


e = eng.FinetuneEngine(synthetic.gnnnet_state_dict(seed=0), 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E, device=dev)
ep = synthetic.test_episode_device(1, dev)
for s in range(E):
    e.load_episode(s, ep)
e.adapt.reset(e.W)
e.prepare_batch()
rs = np.random.RandomState(0)
tables = e.step_tables([[rs.permutation(500)] for _ in range(E)], E)
tables = (tables * (STEPS // len(tables) + 1))[:STEPS]
e.inner_loop(tables[:20]); torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True)
th.start()
runs = []
for alt in range(ALT):
    for tag, knobs in (("A", A), ("B", B)):
        apply(knobs)
        e.inner_loop(tables[:10]); torch.cuda.synchronize()
        t0 = time.time()
        e.inner_loop(tables)
        torch.cuda.synchronize()
        t1 = time.time()
        runs.append((tag, t0, t1))
lib.mft_debug_reset()
stop = True
th.join()
T = np.array([s[0] for s in samples])
P = np.array([[c[0] for c in s[1]] for s in samples])
F = np.array([[c[1] for c in s[1]] for s in samples])
ci = int(np.nanargmax(np.nanmean(P, axis=0)))
res = {"A": [], "B": []}
for tag, t0, t1 in runs:
    sel = (T >= t0 + 0.1) & (T <= t1)
    ms = (t1 - t0) / STEPS * 1e3
    res[tag].append(ms)
    print("%s [%-14s] %.3f ms/step = %.1f episodes/s | %.0f W  %.0f MHz | %.2f J/step" % (
        tag, A if tag == "A" else B, ms, E / (ms * 0.5), np.nanmedian(P[sel, ci]), np.nanmedian(F[sel, ci]),
        np.nanmedian(P[sel, ci]) * ms * 1e-3))
print("mean A %.3f ms  B %.3f ms  (B/A = %.4f; episodes/s quoted for 500 steps per episode, inner loop only)" % (
    np.mean(res["A"]), np.mean(res["B"]), np.mean(res["B"]) / np.mean(res["A"])))
