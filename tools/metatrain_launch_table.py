"""Per-launch table of the meta-training step (BASELINE configs[3]): every C-ABI launcher call of three eager steps timed with HIP
events on its own stream (_lib.LaunchTimer); convolution launches carry their shape and algorithmic TFLOP/s so the launches
that sit far below the matrix rate can be named.  GPU only.  Usage: python tools/metatrain_launch_table.py [n_rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge                                           # noqa: E402

ge.build()
from meta_fine_tuning_amd import _lib, optim, synthetic                # noqa: E402
from meta_fine_tuning_amd.io_utils import model_dict                   # noqa: E402
from meta_fine_tuning_amd.methods.gnnnet import GnnNet                 # noqa: E402


def osz(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def conv_shape(name, a):
    """(n_img, H, W, Cin, Cout, KH, KW, stride, pad) of a convolution-family launcher call, or None."""
    if name in ("mft_conv2d_nhwc", "mft_conv2d_nhwc_ksplit"):
        return a[6:15]
    if name in ("mft_conv2d_dgrad_nhwc", "mft_conv2d_dgrad_nhwc_ksplit"):
        return a[5:14]
    if name in ("mft_conv2d_wgrad_nhwc", "mft_conv2d_wgrad_oihw"):
        return a[5:14]
    return None


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    torch.manual_seed(0)
    model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=5).cuda()
    model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
    model.train()
    model.n_query = 16
    opt = optim.Adam(model.parameters())
    eps = [torch.randn(5, 21, 3, 84, 84, device="cuda") for _ in range(2)]
    for i in range(3):                                                  # warm: workspaces, packed weights
        opt.zero_grad()
        model.set_forward_loss(eps[i % 2]).backward()
        opt.step()
    torch.cuda.synchronize()
    reps = 3
    with _lib.LaunchTimer(keep_args=True) as lt:
        for i in range(reps):
            opt.zero_grad()
            model.set_forward_loss(eps[i % 2]).backward()
            opt.step()
        torch.cuda.synchronize()
        calls = lt.collect(calls=True)
        lt.close()
    per = len(calls) // reps
    rows = []
    for j in range(per):
        name = calls[j][0]
        ms = sum(calls[j + r * per][1] for r in range(reps)) / reps
        assert all(calls[j + r * per][0] == name for r in range(reps))
        shp = conv_shape(name, calls[j][2])
        fl = None
        if shp is not None:
            n, H, W, ci, co, kh, kw, st, pd = (int(v) for v in shp)
            fl = 2.0 * n * osz(H, kh, st, pd) * osz(W, kw, st, pd) * co * kh * kw * ci
        rows.append((j, name, ms, shp, fl))
    total = sum(r[2] for r in rows)
    print("# %d launcher calls per step, %.3f ms of launches per eager step" % (per, total))
    print("# %-4s %-34s %8s %6s  %s" % ("idx", "launcher", "us", "TF", "n H W Cin Cout KH KW s p"))
    for j, name, ms, shp, fl in sorted(rows, key=lambda r: -r[2])[:top]:
        print("%-6d %-34s %8.1f %6s  %s" % (j, name, ms * 1e3, "%.1f" % (fl / ms / 1e9) if fl else "-",
                                          " ".join(str(int(v)) for v in shp) if shp is not None else ""))
    by = {}
    for _, name, ms, _, _ in rows:
        c = by.setdefault(name, [0, 0.0])
        c[0] += 1
        c[1] += ms
    print("# by launcher")
    for name, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("%-40s %4d %8.1f us  %5.1f %%" % (name, n, ms * 1e3, 100 * ms / total))


if __name__ == "__main__":
    main()
