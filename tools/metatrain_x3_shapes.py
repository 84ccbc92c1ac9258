#!/usr/bin/env python3
"""Single-episode (105-image) convolution shapes of the meta-training step: the fp32-MFMA launches the step uses today
(ops.conv2d, K-sliced where that applies) against the split-precision kernels of the frozen trunk (bf16x3 / f16x2 planes of the
same weights; the plane split is timed separately -- in training the weights change every step).  VERDICT r04 next 5 asked for
"bf16x3 for forward and stride-1 data gradients".   GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

ge.build()
from meta_fine_tuning_amd import ops


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


torch.manual_seed(0)
print("%-34s %9s %9s %9s %9s %9s   %s" % ("shape n H Cin Cout k s", "fp32 us", "bf16x3", "f16x2", "split3", "split2", "max|x3 - fp32| / max|out|"))
for n, H, cin, cout, k, s in ((105, 21, 64, 64, 3, 1), (105, 21, 64, 128, 3, 2), (105, 11, 128, 128, 3, 1), (105, 11, 128, 256, 3, 2),
                              (105, 6, 256, 256, 3, 1), (105, 6, 256, 512, 3, 2), (105, 3, 512, 512, 3, 1), (105, 21, 64, 128, 1, 2)):
    pad = 1 if k == 3 else 0
    x = torch.randn(n, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") * (2.0 / (k * k * cin)) ** 0.5
    wp = ops.pack_conv_weight(w)
    t32 = timeit(lambda: ops.conv2d(x, wp, cout, k, k, s, pad))
    ref = ops.conv2d(x, wp, cout, k, k, s, pad)
    w3 = ops.split_weight_x3(wp)
    w2 = ops.split_weight_h2(wp)
    try:
        t3 = timeit(lambda: ops.conv2d_x3(x, w3, cout, k, k, s, pad))
        t2 = timeit(lambda: ops.conv2d_x3(x, w2, cout, k, k, s, pad))
        o3 = ops.conv2d_x3(x, w3, cout, k, k, s, pad)
        err = float((o3 - ref).abs().max() / ref.abs().max())
    except RuntimeError as e:
        t3 = t2 = float("nan")
        err = float("nan")
    ts3 = timeit(lambda: ops.split_weight_x3(wp))
    ts2 = timeit(lambda: ops.split_weight_h2(wp))
    print("%-34s %9.1f %9.1f %9.1f %9.1f %9.1f   %.2e" % ("%d %d %d %d %d %d" % (n, H, cin, cout, k, s), t32, t3, t2, ts3, ts2, err))
