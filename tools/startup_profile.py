#!/usr/bin/env python3
"""Where the fixed cost of a 600-episode evaluation goes (tools; prints one line per phase).
    python tools/startup_profile.py [E]"""
import os, sys, time
t00 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
t_imp = time.perf_counter()
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import engine as eng, functional as Fn, synthetic, ops
import bench

E = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")


def tick(name, t0):
    torch.cuda.synchronize()
    print("%-48s %7.3f s" % (name, time.perf_counter() - t0), flush=True)
    return time.perf_counter()


print("%-48s %7.3f s" % ("python + numpy + torch import", t_imp - t00))
t = time.perf_counter()
torch.zeros(1, device=dev)
t = tick("first device touch (HIP init)", t)
ops._lib.lib()
x = torch.zeros(4, device=dev); ops.softmax_rows(x.view(1, 4))
t = tick("libmft_hip.so load + first launch", t)
state = bench.g9_state()
t = tick("state dict (host)", t)
eps = [synthetic.test_episode_device(7000 + i, dev, 5, 5, 15, 84, 17) for i in range(E)]
t = tick("generate %d episodes on device" % E, t)
free0 = torch.cuda.mem_get_info()[0]
for hints in ("0", "1", "1"):
    os.environ["MFT_SLAB_HINTS"] = hints
    ad = eng.AdaptState(E, dev)
    t = tick("AdaptState(E=%d) hints=%s  [%s]" % (E, hints, (ad.placement or {}).get("source")), t)
    del ad
    torch.cuda.empty_cache()
    t = tick("  free it", t)
os.environ["MFT_SLAB_CANDIDATES"] = "0"
ad = eng.AdaptState(E, dev)
t = tick("AdaptState without placement", t)
del ad
os.environ["MFT_SLAB_CANDIDATES"] = "12"
e = eng.FinetuneEngine(state, n_views=19, fine_tune_epoch=5, episodes_per_batch=E, device=dev)
t = tick("FinetuneEngine build (incl. placement)", t)
e._flip_buffers()
t = tick("  _flip_buffers (second weight slab, final arena)", t)
e._ingest(eps, False)
t = tick("  ingest", t)
e.prepare_batch()
t = tick("  stem cache fill", t)
e.adapt.reset(e.W)
perms = [eng.draw_perms(e.n_total, 5) for _ in range(E)]
t = tick("  reset + draw perms", t)
tabs = e.step_tables(perms, E)
t = tick("  step tables", t)
e.inner_loop(tabs)
t = tick("  inner loop (500 steps)", t)
e.final_scores()
t = tick("  final scores", t)
sc = e.run_batch(eps, defer_final=True, prefetch=eps)
t = tick("second run_batch (defer + prefetch alloc)", t)
sc = e.run_batch(eps, defer_final=True, prefetch=eps)
t = tick("third run_batch (steady state)", t)
print("total %.2f s" % (time.perf_counter() - t00))
