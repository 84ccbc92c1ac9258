"""Child process of tests/test_drivers_gpu.py: one rank of a world_size-2 run launched by torch.distributed.run.
MFT_ONE_DEVICE=1 puts every rank on cuda:0 with a gloo group (a one-GPU box stands in for two GPUs; on a real node the
same drivers use RCCL).  Writes this rank's result as an .npz."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_fine_tuning_amd  # noqa: E402,F401
from meta_fine_tuning_amd import finetune, parallel, synthetic, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    finetune._init_distributed()
    rank, W = parallel.world()
    if a.mode == "finetune":
        from meta_fine_tuning_amd.io_utils import model_dict
        from meta_fine_tuning_amd.methods.gnnnet import GnnNet
        state = synthetic.gnnnet_state_dict(seed=0)
        model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda()
        model.load_state_dict(state)
        state_b = synthetic.gnnnet_state_dict(seed=400)
        accs = finetune.evaluate(model, state, 6, 5, 5, 15, 84, 1, 1, seed0=500, episodes_per_batch=2, verbose=False,
                                 method="all", state_b=state_b, rng_seed=10)
        np.savez(a.out + ".%d.npz" % rank, accs=accs)
    elif a.mode == "train":
        import tempfile
        from meta_fine_tuning_amd import configs
        configs.save_dir = tempfile.mkdtemp()
        torch.manual_seed(0)
        steps = int(os.environ.get("MFT_TEST_TRAIN_STEPS", "2"))
        kk = int(os.environ.get("MFT_TEST_EPISODES_PER_RANK", "1"))          # > 1: k episodes per rank and step in lockstep (round 6)
        extra = ["--episodes_per_rank", str(kk)] if kk > 1 else []
        m = train.main(["--method", "gnnnet", "--model", "ResNet10", "--stop_epoch", "1", "--save_freq", "1"] + extra,
                       n_episode=steps * kk * W + (W - 1), size=84)          # n_episode % W != 0: every rank must still run `steps` steps
        out = {k: v.detach().cpu().numpy() for k, v in m.named_parameters() if k in
               ("fc.0.weight", "gnn.layer_last.fc.weight", "feature.trunk.7.C2.weight", "feature.trunk.0.weight")}
        st = m.__dict__.get("_mft_graph_steps", {}).get("set_forward_loss_lockstep" if kk > 1 else "set_forward_loss")
        out["graphed"] = np.int32(1 if (st is not None and st.graph is not None) else 0)
        np.savez(a.out + ".%d.npz" % rank, **out)
    elif a.mode == "rccl1":
        # ONE rank, backend "nccl" (= RCCL), collectives forced: the calls the 8-GPU drivers make, executed for real on this box's GPU
        import torch.distributed as dist
        assert dist.is_initialized() and dist.get_backend() == "nccl" and W == 1 and parallel.collectives_forced()
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(s_, device="cuda")) for s_ in ((512, 512, 3, 3), (128, 512), (5,), (1179648,))]
        for p_ in ps[:-1]:
            p_.grad = torch.randn_like(p_)
        bucket = parallel.FlatGradBucket(ps)                                   # (the last parameter has no gradient: zeros in the sum, grad stays None)
        want = [None if p_.grad is None else p_.grad.clone() for p_ in ps]
        bucket.flat.fill_(7.0)                                                 # stale bytes in the None parameter's slice must not reach the sum
        bucket.allreduce_mean()                                                # pack -> ncclAllReduce(SUM) over 1 rank -> / 1 -> unpack
        torch.cuda.synchronize()
        ok_ar = all((p_.grad is None and w_ is None) or torch.equal(p_.grad, w_) for p_, w_ in zip(ps, want))
        ok_ar = ok_ar and float(bucket.views[-1].abs().max()) == 0.0
        vals = parallel.gather_episode_values([1.5, 2.5, 99.0], 3, device="cuda")       # ncclAllGather on float64 device buffers
        lin = torch.nn.Linear(8, 4).cuda()
        before = [t.detach().clone() for t in list(lin.parameters()) + list(lin.buffers())]
        parallel.broadcast_parameters(lin, src=0)
        parallel.broadcast_buffers(torch.nn.BatchNorm1d(4).cuda(), src=0)
        torch.cuda.synchronize()
        ok_bc = all(torch.equal(a_, b_) for a_, b_ in zip(before, list(lin.parameters()) + list(lin.buffers())))
        # the sharded evaluation loop end to end on this process group (accuracy gather on the device through RCCL)
        from meta_fine_tuning_amd.io_utils import model_dict
        from meta_fine_tuning_amd.methods.gnnnet import GnnNet
        state = synthetic.gnnnet_state_dict(seed=0)
        model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda()
        model.load_state_dict(state)
        accs = finetune.evaluate(model, state, 3, 5, 5, 15, 84, 1, 1, seed0=500, episodes_per_batch=2, verbose=False, method="gnnnet", rng_seed=10)
        # and one episode-parallel meta-training epoch: AllReduceAdam's flat-bucket all-reduce in front of every outer step
        import tempfile
        from meta_fine_tuning_amd import configs
        configs.save_dir = tempfile.mkdtemp()
        torch.manual_seed(0)
        m = train.main(["--method", "gnnnet", "--model", "ResNet10", "--stop_epoch", "1", "--save_freq", "1"], n_episode=2, size=84)
        try:
            ver = np.asarray(torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            ver = np.asarray([-1])
        np.savez(a.out + ".%d.npz" % rank, ok_allreduce=np.int32(ok_ar), ok_broadcast=np.int32(ok_bc), vals=vals, accs=accs,
                 fc=m.state_dict()["fc.0.weight"].detach().cpu().numpy(), nccl_version=ver)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
