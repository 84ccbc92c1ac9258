#!/usr/bin/env python3
"""episodes/s of the REFERENCE-SHAPED per-episode loop (finetune.py:599-632: one finetune() call per episode, result read at
once) over a LookaheadLoader, against finetune_batched on the same episodes, and against the plain per-episode call.
Usage: lookahead_time.py [n_episodes] [E]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import meta_fine_tuning_amd  # noqa
from meta_fine_tuning_amd import finetune as ft, synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods.gnnnet import GnnNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
E = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda:0"
sd = synthetic.gnnnet_state_dict(seed=0)
model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda()
model.load_state_dict(sd)
ft.params = argparse.Namespace(model='ResNet10', fine_tune_epoch=5)
base = [synthetic.test_episode_device(100 + i, dev) for i in range(min(n, E))]
eps = [base[i % len(base)] for i in range(n)]
y = torch.zeros(5, 20)


class Loader:
    def __len__(self):
        return n

    def __iter__(self):
        for ep in eps:
            yield [(v, y) for v in [u.clone() if False else u for u in ep]]


def loop(loader, limit=None):
    acc_all = []
    for idx, elem in enumerate(loader):
        liz_x = [x for (x, _) in elem]
        scores = ft.finetune(liz_x, None, model, sd, save_it=-1, n_query=15, n_way=5, n_support=5)
        topk_labels = scores.data.topk(1, 1, True, True)[1]
        acc_all.append(float((topk_labels.cpu().numpy()[:, 0] == np.repeat(range(5), 15)).mean()) * 100)
        if limit and idx + 1 >= limit:
            break
    return acc_all


# NOTE: the loader yields the SAME tensor objects for repeated episodes; the registry keys on identity, so give every
# yielded episode its own first-view object
class UniqueLoader(Loader):
    def __iter__(self):
        for ep in eps:
            yield [(ep[0].view_as(ep[0]), y)] + [(v, y) for v in ep[1:]]


np.random.seed(10)
la = ft.LookaheadLoader(UniqueLoader(), "gnnnet", model, sd, None, 5, 5, 5, episodes_per_batch=E)
loop(ft.LookaheadLoader(UniqueLoader(), "gnnnet", model, sd, None, 5, 5, 5, episodes_per_batch=E), limit=E)      # warm
torch.cuda.synchronize(); t0 = time.perf_counter()
loop(la)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("reference-shaped loop over LookaheadLoader(E=%d): %d episodes in %.2f s = %.1f episodes/s" % (E, n, t1 - t0, n / (t1 - t0)))
t0 = time.perf_counter()
ft.finetune_batched(eps, model, sd, 5, 5, 5, E)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("finetune_batched(E=%d):                         %d episodes in %.2f s = %.1f episodes/s" % (E, n, t1 - t0, n / (t1 - t0)))
k = 4
loop(UniqueLoader(), limit=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
loop(UniqueLoader(), limit=k)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("plain per-episode calls (engine of one):        %d episodes in %.2f s = %.1f episodes/s" % (k, t1 - t0, k / (t1 - t0)))
