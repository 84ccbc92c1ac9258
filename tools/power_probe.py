#!/usr/bin/env python3
"""Is the lockstep step power / clock limited?  Samples the GPU's hwmon files (shader clock, memory clock, socket power, temperature)
from a host thread while four workloads run for a few seconds each: the last-block stream alone (HBM-bound), the frozen trunk alone
(matrix-bound), both (the real inner loop) and a pure 3R+3W stream.  If the shader clock under the combined load sits well below the
clock either half runs at alone, the two streams share a power budget and "step time = sum of the streams" follows whatever the
kernels do.   Usage: power_probe.py [E] [seconds per phase]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0


CARDS = []
for card in sorted(glob.glob("/sys/class/drm/card*/device")):
    hw = glob.glob(card + "/hwmon/hwmon*")
    if hw and os.path.exists(hw[0] + "/power1_input"):
        CARDS.append((card, hw[0]))
samples = []
stop = False


def rd(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return float("nan")


def sampler():
    # every GPU of the box is sampled; the one under load is picked afterwards (the visible device is not card0 in general)
    while not stop:
        samples.append((time.time(), [(rd(hw + "/power1_input") * 1e-6, rd(hw + "/freq1_input") * 1e-6, rd(hw + "/freq2_input") * 1e-6,
                                       rd(hw + "/temp2_input") * 1e-3, rd(hw + "/temp3_input") * 1e-3) for _, hw in CARDS]))
        time.sleep(0.02)


th = threading.Thread(target=sampler, daemon=True)
th.start()

import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine as eng, ops, synthetic

dev = "cuda:0"
e = eng.FinetuneEngine(synthetic.gnnnet_state_dict(seed=0), 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E, device=dev)
ep = synthetic.test_episode_device(1, dev)
for s in range(E):
    e.load_episode(s, ep)
e.adapt.reset(e.W)
e.prepare_batch()
rs = np.random.RandomState(0)
perms = [[rs.permutation(500)] for _ in range(E)]
tables = e.step_tables(perms, E)
e.inner_loop(tables[:10]); torch.cuda.synchronize()

orig = {k: getattr(ops, k) for k in ("conv2d_x3_bnstats", "conv2d_x3", "bn_apply", "bn_combine_moments")}
orig_gather = e.stem.gather
phases = []


def phase(tag, fn):
    torch.cuda.synchronize()
    time.sleep(1.0)
    t0 = time.time()
    n = 0
    while time.time() - t0 < SECS:
        fn()
        torch.cuda.synchronize()
        n += 1
    t1 = time.time()
    phases.append((tag, t0, t1, n))
    print("%-34s %d repetitions in %.2f s = %.3f ms each" % (tag, n, t1 - t0, (t1 - t0) / n * 1e3), flush=True)


def no_trunk():
    ops.conv2d_x3_bnstats = lambda x, w3, Cout, KH, KW, stride, pad, ipg, out, ws, mean, rstd, **kw: (out, mean, rstd)
    ops.conv2d_x3 = lambda x, w3, Cout, KH, KW, stride, pad, out=None: out
    ops.bn_apply = lambda x2d, C, rpg, ng, mean, rstd, g, b, act=0, res=None, res_bn=None, out=None, **kw: out
    ops.bn_combine_moments = lambda *a, **kw: None
    e.stem.gather = lambda idx, n, m, s, g, b, ipg, out, planes=None: out


def restore():
    for k, f in orig.items():
        setattr(ops, k, f)
    e.stem.gather = orig_gather


idx_dev = [torch.from_numpy(t[1]).to(dev) for t in tables[:20]]
k = tables[0][0]
phases.append(("idle", time.time(), time.time() + 1.5, 0))
time.sleep(1.5)
no_trunk()
phase("last-block stream alone (20 steps)", lambda: e.inner_loop(tables[:20]))
orig_wgrad = ops.conv2d_wgrad_adam
ops.conv2d_wgrad_adam = lambda *a, **kw: None
phase("  ... without weight-gradient + Adam", lambda: e.inner_loop(tables[:20]))
ops.conv2d_wgrad_adam = orig_wgrad
restore()
phase("trunk alone (20 steps)", lambda: [e.trunk_step(i, k, 0) for i in idx_dev])
phase("both: the inner loop (20 steps)", lambda: e.inner_loop(tables[:20]))
n_probe = 1 << 28
scratch = torch.empty(3 * n_probe, device=dev)
lib = _lib.lib()
phase("pure 3R+3W stream (3.2 GB arrays)", lambda: lib.mft_stream_probe(ops._p(scratch), ops._p(scratch[n_probe:]), ops._p(scratch[2 * n_probe:]),
                                                                         n_probe, ops._stream()))
stop = True
th.join()


T = np.array([r[0] for r in samples])
V = np.array([r[1] for r in samples])                      # [sample][card][power, sclk, mclk, t_junction, t_mem]
busy = (T >= phases[1][1]) & (T <= phases[-1][2])
ci = int(np.nanargmax(np.nanmean(V[busy, :, 0], axis=0)))
cap = rd(CARDS[ci][1] + "/power1_cap") * 1e-6
print("\nsensor: %s (power cap %.0f W)" % (CARDS[ci][0], cap))
print("%-36s %9s %9s %16s %10s %10s %9s" % ("phase", "ms each", "W median", "sclk MHz [min]", "mclk MHz", "T junction", "J each"))
for tag, t0, t1, n in phases:
    sel = (T >= t0 + 0.3) & (T <= t1)
    v = V[sel, ci]
    ms = (t1 - t0) / n * 1e3 if n else float("nan")
    print("%-36s %9.2f %9.0f %9.0f [%4.0f] %10.0f %10.0f %9.3f" % (tag, ms, np.nanmedian(v[:, 0]), np.nanmedian(v[:, 1]), np.nanmin(v[:, 1]),
                                                             np.nanmedian(v[:, 2]), np.nanmedian(v[:, 3]), np.nanmedian(v[:, 0]) * ms * 1e-3))
