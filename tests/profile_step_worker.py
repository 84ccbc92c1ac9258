"""Child process of tests/test_metatrain_gpu.py::test_meta_training_step_issues_no_torch_device_ops: one eager meta-training step
under the torch profiler; prints ONE json line {"n_dev": device kernels seen, "aten": [names of ATen device kernels], "nbt_ok": the
BatchNorm counters moved by exactly one}.  Kept out of the pytest process: the profiler's tracing thread (kineto / roctracer) was seen
to abort the whole process some seconds AFTER the profiled region, once in about thirty runs -- the result line is printed before."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_fine_tuning_amd  # noqa: E402,F401
from meta_fine_tuning_amd import optim, synthetic  # noqa: E402
from meta_fine_tuning_amd.io_utils import model_dict  # noqa: E402
from meta_fine_tuning_amd.methods.gnnnet import GnnNet  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile
    model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda()
    model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
    model.train()
    model.n_query = 16
    opt = optim.Adam(model.parameters())
    x = synthetic.train_episode(5, 5, 5, 16, 84).cuda()
    one = torch.ones((), device="cuda")
    for _ in range(2):
        opt.zero_grad()
        model.set_forward_loss(x).backward(one)
        opt.step()
    torch.cuda.synchronize()
    nbt0 = int(model.feature.trunk[1].num_batches_tracked)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        opt.zero_grad()
        model.set_forward_loss(x).backward(one)
        opt.step()
        torch.cuda.synchronize()
    dev_kernels = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    aten = sorted({n for n in dev_kernels if "at::native" in n or n.startswith("void at::")})
    nbt_ok = int(model.feature.trunk[1].num_batches_tracked) == nbt0 + 1 == int(model.feature.trunk[7].BN2.num_batches_tracked)
    print("RESULT " + json.dumps({"n_dev": len(dev_kernels), "aten": aten, "nbt_ok": bool(nbt_ok)}), flush=True)


if __name__ == "__main__":
    main()
