#!/usr/bin/env python3
"""Known-traffic launches for calibrating FETCH_SIZE / WRITE_SIZE on this rocprofv3 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE
counts 128-B requests as 64 B; WRITE_SIZE uncalibrated).  Each launch touches exactly N = 1 GiB per array:
  mft_adam_step  : reads p,g,m,v (4 GiB), writes p,m,v (3 GiB)     [flat float4 stream, same access width as the fused epilogue]
  torch fill     : writes 1 GiB
  torch sum      : reads 1 GiB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import ops
n = 1 << 28
p = torch.zeros(n, device="cuda"); g = torch.zeros(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
torch.cuda.synchronize()
for i in range(3):
    ops.adam_step(p, g, m, v, i + 1)
    p.fill_(1.0)
    s = p.sum()
torch.cuda.synchronize()
print("calib done", float(s))
