import sys, torch
sys.path.insert(0, "/root/repo")
import meta_fine_tuning_amd
from meta_fine_tuning_amd import ops, _lib
n = 8192
x = torch.randn(n, 84, 84, 3, device="cuda")
w = ops.pack_conv_weight(torch.randn(64, 3, 7, 7, device="cuda") * 0.1)
out = torch.empty(n, 42, 42, 64, device="cuda")
for mode in (2000, 2001):
    _lib.lib().mft_debug_set_conv_tile(mode)
    ops.conv2d(x, w, 64, 7, 7, 2, 3, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.conv2d(x, w, 64, 7, 7, 2, 3, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 5
    print("stem mode %d: %.0f us for %d images = %.1f TF, out write %.2f TB/s" % (mode, us, n, n * 1764 * 64 * 147 * 2 / us / 1e6, out.numel() * 4 / us / 1e6))
    if mode == 2000: ref = out.clone()
print("max diff fast vs generic:", float((out - ref).abs().max()))
