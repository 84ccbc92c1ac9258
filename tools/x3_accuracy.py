"""Error of the three fp32-accurate convolution forms against float64 on the frozen-trunk shapes, and their launch times:
fp32 MFMA (conv_igemm), bf16x3 (three bf16 pieces, six products), f16x2 (two fp16 pieces, three products).
    gpurun -- python tools/x3_accuracy.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_fine_tuning_amd  # noqa: F401,E402
from meta_fine_tuning_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [("trunk.4.C1", 64, 64, 3, 1, 1, 21), ("trunk.5.C1", 64, 128, 3, 2, 1, 21), ("trunk.5.C2", 128, 128, 3, 1, 1, 11),
          ("trunk.5.shortcut", 64, 128, 1, 2, 0, 21), ("trunk.6.C1", 128, 256, 3, 2, 1, 11), ("trunk.6.C2", 256, 256, 3, 1, 1, 6),
          ("trunk.6.shortcut", 128, 256, 1, 2, 0, 11)]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    lib = _lib.lib()
    print("%-18s %6s | %-26s | %-26s | %-26s" % ("layer", "K", "fp32 MFMA  max / rms  us", "bf16x3  max / rms  us", "f16x2  max / rms  us"))
    for name, Cin, Cout, k, stride, pad, H in SHAPES:
        g = torch.Generator().manual_seed(5)
        n_acc, n_time = 16, 640
        x = torch.relu(torch.randn((n_time, Cin, H, H), generator=g) + 0.3)
        w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cout)) ** 0.5
        ref = F.conv2d(x[:n_acc].double(), w.double(), None, stride, pad).permute(0, 2, 3, 1)
        xg = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wpk = ops.pack_conv_weight(w.to(DEV))
        w3, w2 = ops.split_weight_x3(wpk), ops.split_weight_h2(wpk)
        OH = (H + 2 * pad - k) // stride + 1
        out = torch.empty((n_time, OH, OH, Cout), device=DEV)
        scale = float(ref.abs().max())
        res = []

        def f32():
            assert lib.mft_conv2d_nhwc(ops._p(xg), Cin, ops._p(wpk), None, ops._p(out), Cout, n_time, H, H, Cin, Cout, k, k, stride, pad, 0, 0,
                                       ops._stream()) == 0
        for fn in (f32, lambda: ops.conv2d_x3(xg, w3, Cout, k, k, stride, pad, out=out), lambda: ops.conv2d_x3(xg, w2, Cout, k, k, stride, pad, out=out)):
            us = timed(fn)
            d = out[:n_acc].double().cpu() - ref
            res.append((float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale, us))
        print("%-18s %6d | %s" % (name, k * k * Cin, " | ".join("%9.2e %9.2e %6.1f" % r for r in res)))


if __name__ == "__main__":
    main()
