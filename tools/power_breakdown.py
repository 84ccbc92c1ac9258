#!/usr/bin/env python3
"""Socket power as a profiler: runs single kernels of the inner loop (and ablated forms of them: no matrix work, no Adam arithmetic,
no operand split ...) in a loop for a few seconds each and reports time per launch, median socket power, shader clock and ENERGY per
launch (power above the idle floor x time).  The step is power-limited (tools/power_probe.py): what counts is joules, not busy
pipes.   Usage: power_breakdown.py [E] [seconds per case]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0

CARDS = []
for card in sorted(glob.glob("/sys/class/drm/card*/device")):
    hw = glob.glob(card + "/hwmon/hwmon*")
    if hw and os.path.exists(hw[0] + "/power1_input"):
        CARDS.append((card, hw[0] + "/power1_input", hw[0] + "/freq1_input"))
samples = []
stop = False


def rd(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return float("nan")


def sampler():
    while not stop:
        samples.append((time.time(), [(rd(pw) * 1e-6, rd(fq) * 1e-6) for _, pw, fq in CARDS]))
        time.sleep(0.02)


import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

dev = "cuda:0"
lib = _lib.lib()
torch.zeros(1, device=dev)
th = threading.Thread(target=sampler, daemon=True)
th.start()
cases = []


def case(tag, fn, bytes_=0.0):
    fn(); torch.cuda.synchronize()
    time.sleep(0.6)
    t0 = time.time()
    n = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.time() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    t1 = time.time()
    cases.append((tag, t0, t1, a.elapsed_time(b) * 1e3 / n, bytes_))


g = torch.Generator(device=dev); g.manual_seed(1)
time.sleep(1.5)
idle_t = (time.time() - 1.2, time.time())

# ---- the dominant launch: weight gradient + Adam of trunk.7.C2
Cin = Cout = 512
x = torch.randn(E * 5, 3, 3, Cin, device=dev, generator=g)
dy = torch.randn(E * 5, 3, 3, Cout, device=dev, generator=g) * 1e-3
w = torch.randn(E, Cout, 9 * Cin, device=dev, generator=g) * 0.02
m, v = torch.zeros_like(w), torch.zeros_like(w)
nb = 24.0 * w.numel()
for pol, tag in ((7, "wgrad+Adam C2 (default)"), (15, "  without the matrix work"), (23, "  without the Adam arithmetic"), (31, "  neither (operands + stream)")):
    lib.mft_debug_set_conv_tile(9000 + pol)
    case(tag, lambda: ops.conv2d_wgrad_adam(x, dy, w, m, v, Cout, 3, 3, 1, 1, 5, 5), nb)
lib.mft_debug_set_conv_tile(9003)
case("  exact division / sqrt epilogue", lambda: ops.conv2d_wgrad_adam(x, dy, w, m, v, Cout, 3, 3, 1, 1, 5, 5), nb)
lib.mft_debug_reset()
n_probe = w.numel()
case("pure 3R+3W stream, same bytes", lambda: lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n_probe, ops._stream()), nb)

# ---- the frozen trunk's bf16x3 convolutions (five shapes per step) and their ablations
LAYERS = [("trunk.4.C1", 64, 64, 3, 1, 1, 21), ("trunk.5.C1", 64, 128, 3, 2, 1, 21), ("trunk.5.C2", 128, 128, 3, 1, 1, 11),
          ("trunk.6.C1", 128, 256, 3, 2, 1, 11), ("trunk.6.C2", 256, 256, 3, 1, 1, 6)]
data = []
for (name, cin, cout, k, s, p, H) in LAYERS:
    xx = torch.randn(E * 5, H, H, cin, device=dev, generator=g)
    w3 = ops.split_weight_x3(ops.pack_conv_weight(torch.randn(cout, cin, k, k, device=dev, generator=g) * 0.05))
    data.append((xx, w3, cout, k, s, p, ops.conv2d_x3(xx, w3, cout, k, k, s, p)))


def trunk_convs():
    for (xx, w3, cout, k, s, p, o) in data:
        ops.conv2d_x3(xx, w3, cout, k, k, s, p, out=o)


for knob, tag in ((None, "trunk convolutions x5 (default)"), (202, "  without the MFMAs"), (201, "  without the operand split"),
                  (216, "  stage only (loads, split, LDS stores)"), (212, "  multiply only (fragment reads, MFMAs)"),
                  (228, "  barriers only")):
    lib.mft_debug_reset()
    if knob is not None:
        lib.mft_debug_set_x3_tile(knob)
    case(tag, trunk_convs)
lib.mft_debug_reset()
stop = True
th.join()

# the card under load = the one with the highest mean power over the cases
T = np.array([s[0] for s in samples])
P = np.array([[c[0] for c in s[1]] for s in samples])
F = np.array([[c[1] for c in s[1]] for s in samples])
busy = (T >= cases[0][1]) & (T <= cases[-1][2])
ci = int(np.nanargmax(np.nanmean(P[busy], axis=0)))
idle = float(np.nanmedian(P[(T >= idle_t[0]) & (T <= idle_t[1]), ci]))
print("sensor: %s; idle floor %.0f W (clocks up, no kernel)" % (CARDS[ci][0], idle))
print("%-44s %9s %7s %6s %9s %9s %8s" % ("case", "us/launch", "W", "MHz", "J/launch", "dyn J", "TB/s"))
for tag, t0, t1, us, nbytes in cases:
    sel = (T >= t0 + 0.4) & (T <= t1)
    pw, fq = float(np.nanmedian(P[sel, ci])), float(np.nanmedian(F[sel, ci]))
    print("%-44s %9.1f %7.0f %6.0f %9.3f %9.3f %8s" % (tag, us, pw, fq, pw * us * 1e-6, (pw - idle) * us * 1e-6,
                                                  "%.2f" % (nbytes / us / 1e6) if nbytes else ""))
