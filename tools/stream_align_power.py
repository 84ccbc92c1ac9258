#!/usr/bin/env python3
"""Does the relative placement of the w / m / v arrays change what a 3R+3W stream costs?  (tools/power_probe.py saw 857-869 W for a pure
stream over three 1 GiB arrays carved from one allocation, tools/power_breakdown.py 995-999 W for the same kernel over three separately
allocated 1.2 GB arrays, at the same 5.4 TB/s.)  Runs mft_stream_probe over three arrays placed at chosen distances inside one big
buffer and reports rate, socket power and energy per GB.   Usage: stream_align_power.py [seconds per case]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, ops

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
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


N2 = 1 << 28
NC = 128 * 512 * 4608                       # trunk.7.C2 x 128 episodes
NS = 128 * 3673088 // 1024 * 1024           # one whole parameter slab at E = 128
early = [torch.zeros(NC, device=dev) for _ in range(3)]           # three separate allocations made FIRST in the process
early_s = [torch.zeros(NS, device=dev) for _ in range(3)]
big = torch.empty(3 * (1 << 29) + (1 << 26), device=dev)          # 6.3 GiB of floats
big.zero_()
th = threading.Thread(target=sampler, daemon=True)
th.start()
cases = []


def run(tag, n, offs, tensors=None):
    if tensors is None:
        tensors = [big[o:o + n] for o in offs]
    w, m, v = tensors
    lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
    torch.cuda.synchronize()
    time.sleep(0.5)
    t0 = time.time()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    k = 0
    while time.time() - t0 < SECS:
        for _ in range(50):
            lib.mft_stream_probe(ops._p(w), ops._p(m), ops._p(v), n, ops._stream())
        k += 50
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    cases.append((tag, t0, time.time(), a.elapsed_time(b) * 1e3 / k, 24.0 * n))


run("C2-sized, three allocations made first", NC, None, early)
run("slab-sized, three allocations made first", NS, None, early_s)
run("2^28 floats each, back to back (1 GiB apart)", N2, (0, N2, 2 * N2))
run("2^28 floats each, 1 GiB + 16 MiB apart", N2, (0, N2 + (1 << 22), 2 * (N2 + (1 << 22))))
run("2^28 floats each, 1 GiB + 4 KiB apart", N2, (0, N2 + 1024, 2 * (N2 + 1024)))
run("C2-sized (1.21 GB) back to back", NC, (0, NC, 2 * NC))
run("C2-sized, 2 GiB apart", NC, (0, 1 << 29, 2 << 29))
run("C2-sized, 1.5 GiB apart", NC, (0, 3 << 27, 6 << 27))
sep = [torch.zeros(NC, device=dev) for _ in range(3)]
run("C2-sized, three separate allocations", NC, None, sep)
run("C2-sized, three allocations made first (again)", NC, None, early)
run("C2-sized, three separate allocations (again)", NC, None, sep)
del sep
run("slab-sized (1.88 GB) back to back", NS, (0, NS, 2 * NS))
run("slab-sized, 2 GiB apart", NS, (0, 1 << 29, 2 << 29))
sep_s = [torch.zeros(NS, device=dev) for _ in range(3)]
run("slab-sized, three separate allocations", NS, None, sep_s)
one = torch.zeros(3 * NC, device=dev)
run("C2-sized, one fresh allocation of exactly 3n", NC, None, [one[:NC], one[NC:2 * NC], one[2 * NC:]])
stop = True
th.join()
T = np.array([s[0] for s in samples])
P = np.array([[c[0] for c in s[1]] for s in samples])
F = np.array([[c[1] for c in s[1]] for s in samples])
ci = int(np.nanargmax(np.nanmean(P, axis=0)))
print("%-48s %9s %8s %7s %6s %8s" % ("placement of w / m / v", "us/pass", "TB/s", "W", "MHz", "J/GB"))
for tag, t0, t1, us, nb in cases:
    sel = (T >= t0 + 0.3) & (T <= t1)
    pw = float(np.nanmedian(P[sel, ci]))
    print("%-48s %9.1f %8.2f %7.0f %6.0f %8.4f" % (tag, us, nb / us / 1e6, pw, float(np.nanmedian(F[sel, ci])), pw * us * 1e-6 / (nb / 1e9)))
