#!/usr/bin/env python3
"""Per-launch breakdown of one lockstep inner step (single stream, HIP events around every C-ABI launch).

Wraps the ctypes handle of libmft_hip.so with a proxy that records an event pair per call and aggregates by
(entry point, shape signature).  Usage: python tools/step_breakdown.py [E] [steps]
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine as eng, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def sig(name, args):
    """Short shape signature from the integer arguments of a launch."""
    ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool)]
    return name + "(" + ",".join(str(i) for i in ints[:12]) + ")"


class Proxy:
    def __init__(self, h):
        self._h = h
        self.on = False
        self.rec = []

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name.endswith("_floats") or name in ("mft_version", "mft_debug_set_conv_tile"):
            return fn

        def call(*args):
            if not self.on:
                return fn(*args)
            s = torch.cuda.current_stream()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            rc = fn(*args)
            b.record(s)
            self.rec.append((sig(name, args), a, b))
            return rc
        return call


def main():
    real = _lib.lib()
    px = Proxy(real)
    _lib._lib = px
    dev = "cuda:0"
    state = synthetic.gnnnet_state_dict(seed=0)
    e = eng.FinetuneEngine(state, 5, 5, 15, 84, n_views=19, fine_tune_epoch=1, episodes_per_batch=E, device=dev,
                           pipeline=False)
    ep = synthetic.test_episode_device(1, dev)
    for slot in range(E):
        e.load_episode(slot, ep)
    e.adapt.reset(e.W)
    if hasattr(e, "prepare_batch"):
        px.on = True
        e.prepare_batch()
        torch.cuda.synchronize()
        agg0 = collections.OrderedDict()
        for k, a, b in px.rec:
            agg0.setdefault(k, [0, 0.0])
            agg0[k][0] += 1
            agg0[k][1] += a.elapsed_time(b)
        print("== prepare_batch (once per batch of %d episodes) ==" % E)
        for k, (n, ms) in agg0.items():
            print("%-110s %4d  %9.1f us" % (k[:110], n, ms * 1e3))
        px.rec = []
        px.on = False
    perms = [eng.draw_perms(e.n_total, 1, np.random.RandomState(i)) for i in range(E)]
    tables = e.step_tables(perms, E)[:STEPS + 2]
    idx = [torch.from_numpy(t[1]).to(dev) for t in tables]
    lab = [torch.from_numpy(t[2]).to(dev) for t in tables]
    for t in range(2):
        e.inner_step(idx[t], lab[t], 5)
    torch.cuda.synchronize()
    px.on = True
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    for t in range(2, STEPS + 2):
        e.inner_step(idx[t], lab[t], 5)
    w1.record()
    torch.cuda.synchronize()
    px.on = False
    agg = collections.OrderedDict()
    for k, a, b in px.rec:
        agg.setdefault(k, [0, 0.0])
        agg[k][0] += 1
        agg[k][1] += a.elapsed_time(b)
    tot = sum(v[1] for v in agg.values())
    print("== inner step, E=%d: %.1f us of launches per step, wall %.1f us per step (events inflate wall) ==" %
          (E, tot * 1e3 / STEPS, w0.elapsed_time(w1) * 1e3 / STEPS))
    for k, (n, ms) in agg.items():
        print("%-110s %4.1f/step  %9.1f us  %5.1f%%" % (k[:110], n / STEPS, ms * 1e3 / n, 100 * ms / tot))


if __name__ == "__main__":
    main()
