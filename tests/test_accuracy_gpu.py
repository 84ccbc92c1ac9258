"""600-episode accuracy parity (BASELINE.json north_star: +-0.2 % on the 600-episode mean) against per-episode accuracies
the REFERENCE's own finetune() produced on the same structured synthetic episodes, weights (seeded backbone + the
meta-trained head fixture g9_head.npz), and numpy permutation stream (oracle/make_golden_g9.py)."""
import os

import numpy as np
import pytest
import torch

import meta_fine_tuning_amd  # noqa: F401
from meta_fine_tuning_amd import engine as eng
from meta_fine_tuning_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _state(golden_dir, g):
    sd = synthetic.gnnnet_state_dict(seed=int(g["seed_sd"]))
    hz = np.load(os.path.join(golden_dir, "g9_head.npz"))
    for k in hz.files:
        sd[k] = torch.from_numpy(hz[k])
    return sd


def _run(golden_dir, tag, batch):
    g = np.load(os.path.join(golden_dir, "g9_accuracy.npz"))
    if "acc_" + tag not in g.files:
        pytest.skip("golden config %s not generated" % tag)
    E_ep, G, n = [int(v) for v in g["cfg_" + tag]]
    ref = g["acc_" + tag]
    n = len(ref)
    sd = _state(golden_dir, g)
    e = eng.FinetuneEngine(sd, 5, 5, 15, 84, n_views=2 + G, fine_tune_epoch=E_ep, episodes_per_batch=batch, device=DEV)
    y = np.repeat(np.arange(5), 15)
    accs, chk = [], []
    np.random.seed(10)                                   # finetune.py:425; permutations are then drawn episode by episode
    from concurrent.futures import ThreadPoolExecutor

    def make(j):
        return synthetic.test_episode(int(g["ep_seed0"]) + j, 5, 5, 15, 84, gen_examples=G, noise=float(g["noise"]))
    with ThreadPoolExecutor(max_workers=8) as ex:         # the numpy episode generator is the slow part (~1 s per 19-view episode)
        for i in range(0, n, batch):
            eps = list(ex.map(make, range(i, min(i + batch, n))))
            sc = e.run_batch(eps).cpu().numpy()
            for s in sc:
                accs.append(float((s.argmax(1) == y).mean() * 100.0))
                chk.append(float(s[:, 0].astype(np.float64).sum()))
    e.close()
    return np.array(accs), np.array(chk), ref, g["chk_" + tag]


def test_g9_accuracy_600_episodes_short_config(golden_dir):
    """fine_tune_epoch=1, gen_examples=2 (20 Adam steps per episode), 600 episodes."""
    accs, chk, ref, ref_chk = _run(golden_dir, "A", 50)
    assert len(accs) == 600
    assert abs(accs.mean() - ref.mean()) <= 0.2, (accs.mean(), ref.mean())
    # Per episode no two fp32 implementations agree exactly: Adam's first steps move every weight by lr*sign(g), so a
    # rounding-level change of a near-zero gradient flips a 0.01 weight move.  The reference's own fp32 run
    # differs from its fp64 run by one query in ~30 % of these episodes; rare episodes with a large set of noise-level
    # gradients jump by several queries.  Distributional bounds: most episodes identical, 99 % within two queries.
    d = np.abs(accs - ref)
    assert np.mean(d < 1e-6) >= 0.6, np.mean(d < 1e-6)
    assert np.percentile(d, 99) <= 2.7 + 1e-6, np.percentile(d, 99)


def test_g9_accuracy_full_config(golden_dir):
    """BASELINE configs[1]: fine_tune_epoch=5, gen_examples=17 (500 Adam steps per episode)."""
    accs, chk, ref, ref_chk = _run(golden_dir, "B", 64)
    n = len(ref)                                 # 600 once oracle/make_golden_g9.py --nB 600 has finished (it saves as it goes)
    # the north_star bar, +-0.2 % on the 600-episode mean; proportionally wider while the fixture holds fewer episodes
    tol = 0.2 if n >= 600 else 0.2 * (600.0 / n) ** 0.5 + 0.1
    assert abs(accs.mean() - ref.mean()) <= tol, (n, accs.mean(), ref.mean())
    # 500 chaotic Adam steps per episode: per-episode identity is not a meaningful bar (the result of ONE episode even depends on
    # its slot in the batch through the summation order of the per-tile BatchNorm statistics); distributional bounds instead:
    # 90 % of the episodes within two queries, 99 % within six, at most one episode in 200 further than eight queries
    d = np.abs(accs - ref)
    assert np.percentile(d, 90) <= 2.7 + 1e-6 and np.percentile(d, 99) <= 8.0 + 1e-6, (np.percentile(d, 90), np.percentile(d, 99))
    assert np.mean(d > 10.7) <= 0.005, (np.mean(d > 10.7), d.max())
    assert d.max() <= 16.0 + 1e-6, d.max()              # hard cap: no episode further than 12 of its 75 queries from the reference
