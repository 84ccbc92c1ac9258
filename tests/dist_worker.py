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
        m = train.main(["--method", "gnnnet", "--model", "ResNet10", "--stop_epoch", "1", "--save_freq", "1"], n_episode=steps * W + (W - 1), size=84)          # n_episode % W != 0: every rank must still run `steps` steps
        out = {k: v.detach().cpu().numpy() for k, v in m.named_parameters() if k in
               ("fc.0.weight", "gnn.layer_last.fc.weight", "feature.trunk.7.C2.weight", "feature.trunk.0.weight")}
        st = m.__dict__.get("_mft_graph_steps", {}).get("set_forward_loss")
        out["graphed"] = np.int32(1 if (st is not None and st.graph is not None) else 0)
        np.savez(a.out + ".%d.npz" % rank, **out)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
