#!/usr/bin/env python3
"""How far apart do EQUALLY VALID fp32 implementations of the inner loop land after 105 / 500 Adam steps?  Runs golden G4b's setting
(one episode, the reference's index order) under kernel variants that differ only in fp32 rounding (K-summation order of the trunk
convolutions, exact vs hardware division in the Adam epilogue, 64x64 vs 32x128 weight-gradient tiles) and prints the last-block weight
norms beside the reference's own fp32 and fp64 runs.  The spread is the envelope a trajectory test can ask for."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import _lib, engine as eng, synthetic

g = np.load(os.path.join(ROOT, "tests", "golden", "g4b_inner_loop_long.npz"))
lib = _lib.lib()
sd = synthetic.resnet10_state_dict(seed=9)
views = synthetic.test_episode(31, 5, 5, 15, 84, gen_examples=17)
full = {"feature." + k: v for k, v in sd.items()}
full.update(synthetic.gnn_head_state_dict(seed=1))
order = g["order"]
keys = (("trunk.7.C1.weight", "wn_c1"), ("trunk.7.C2.weight", "wn_c2"), ("trunk.7.shortcut.weight", "wn_sc"))
for tag in (105, 500):
    print("after %d steps: reference fp32 %s | fp64 %s" % (tag, ["%.3f" % float(g["%s_s%d_f32" % (k, tag)]) for _, k in keys],
                                                         ["%.3f" % float(g["%s_s%d_f64" % (k, tag)]) for _, k in keys]))
    for name, knobs in (("default", ()), ("per-tap trunk convolutions", ("x90",)), ("exact Adam epilogue", ("c9003",)),
                        ("per-tap + exact epilogue", ("x90", "c9003")), ("64x64 weight-gradient tiles", ("c9500",)),
                        ("fp32-MFMA per-episode kernels", ("c8000",)), ("unfused last block", ("env",))):
        lib.mft_debug_reset()
        for k in knobs:
            if k.startswith("c"):
                lib.mft_debug_set_conv_tile(int(k[1:]))
            elif k.startswith("x"):
                lib.mft_debug_set_x3_tile(int(k[1:]))
        from meta_fine_tuning_amd import functional as Fn
        Fn.FUSED_LAST_BLOCK = knobs != ("env",)
        e = eng.FinetuneEngine(full, n_views=19, fine_tune_epoch=5, episodes_per_batch=1, device="cuda:0")
        e._ingest([views], False)
        e.adapt.reset(e.W)
        e.prepare_batch()
        perms = [[order[ep * 500:(ep + 1) * 500] for ep in range(5)]]
        e.inner_loop(e.step_tables(perms, 1)[:tag])
        torch.cuda.synchronize()
        w = e.adapt.w.export(0)
        print("  %-34s %s" % (name, ["%.3f" % float(w[k].norm()) for k, _ in keys]))
        e.close()
    Fn.FUSED_LAST_BLOCK = True
lib.mft_debug_reset()
