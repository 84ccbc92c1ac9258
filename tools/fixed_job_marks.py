"""Phase marks of the fixed 600-episode job (bench.strong_scaling_leg's call of finetune.evaluate) in a fresh process: where the
wall time beyond the steady-state batches goes.  GPU only.  Usage: python tools/fixed_job_marks.py [e_max] [n_episodes]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge                                           # noqa: E402

ge.build()
import bench                                                           # noqa: E402
from meta_fine_tuning_amd import finetune as ft                        # noqa: E402
from meta_fine_tuning_amd.io_utils import model_dict                   # noqa: E402
from meta_fine_tuning_amd.methods.gnnnet import GnnNet                 # noqa: E402

e_max = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
state = bench.g9_state()
model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=5).cuda()
model.load_state_dict(state)
torch.cuda.synchronize()
np.random.seed(10)
tm = {}
t0 = time.perf_counter()
accs = ft.evaluate(model, state, n, 5, 5, 15, 84, 17, 5, seed0=7000, episodes_per_batch=e_max, verbose=False, method="gnnnet",
                   rng_seed=10, device_episodes=True, balance=True, timings=tm)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("e_max %d: %d episodes in %.3f s = %.2f episodes/s; batches %s x %s; engine ready %.3f s; acc %.2f"
      % (e_max, n, dt, n / dt, tm["batches"], tm["episodes_per_batch"], tm.get("engine_ready_s", -1), accs.mean()))
for name, t in tm["marks"]:
    print("  %8.3f  %s" % (t, name))
eng = ft._ENGINES.entries[0]["engine"] if ft._ENGINES.entries else None
if eng is not None:
    print("  fused last loop:", eng.fused_last_loop, "placement:", (eng.adapt.placement or {}).get("source"))
