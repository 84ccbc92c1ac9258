"""Drivers on a real MI355X: engine cache, the reference-shaped loop behind LookaheadLoader, the CLI surface of
finetune.py / finetune_50.py (--method gnnnet | baseline | all, --freeze_backbone, --n_shot 50), the 20-/50-shot configs
against the reference's own outputs (G13 / G14), and the multi-GPU wiring of both drivers (2 ranks, fresh child processes)."""
import argparse
import copy
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import meta_fine_tuning_amd  # noqa: F401
from meta_fine_tuning_amd import engine as eng
from meta_fine_tuning_amd import finetune as ft
from meta_fine_tuning_amd import finetune_50 as ft50
from meta_fine_tuning_amd import synthetic
from meta_fine_tuning_amd.io_utils import model_dict
from meta_fine_tuning_amd.methods import gnnnet_copy
from meta_fine_tuning_amd.methods.gnnnet import GnnNet
from oracle import mft_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _model(sd, n_support=5, cls=GnnNet):
    m = cls(model_dict['ResNet10'], n_way=5, n_support=n_support)
    m.load_state_dict(sd)
    return m.cuda()


# ------------------------------------------------------------------------------------------------ engine cache

def test_engine_cache_keys_on_checkpoint_content():
    ft._ENGINES.clear()
    ft.params = argparse.Namespace(model='ResNet10', fine_tune_epoch=1)
    sd = synthetic.gnnnet_state_dict(seed=13)
    model = _model(sd)
    liz = synthetic.test_episode(41, 5, 5, 15, 84, gen_examples=0)
    np.random.seed(10)
    a = ft.finetune(liz, None, model, sd, None).cpu().numpy()
    assert len(ft._ENGINES.entries) == 1
    e0 = ft._ENGINES.entries[0]["engine"]
    # a deep copy of the state per episode (what the reference does inside finetune, and oracle/make_golden.py outside):
    # same bytes -> same engine, no rebuild
    np.random.seed(10)
    b = ft.finetune(liz, None, model, copy.deepcopy(sd), None).cpu().numpy()
    assert len(ft._ENGINES.entries) == 1 and ft._ENGINES.entries[0]["engine"] is e0
    assert np.array_equal(a, b)
    # the SAME dict object with different weights (in-place update): must not reuse the packed backbone
    with torch.no_grad():
        sd["feature.trunk.7.C2.weight"].mul_(0.5)
    np.random.seed(10)
    c = ft.finetune(liz, None, model, sd, None).cpu().numpy()
    assert len(ft._ENGINES.entries) == 2 and np.abs(c - a).max() > 1e-4
    # a different checkpoint allocated where a freed one lived cannot alias it: entries pin their state dicts
    assert all(ent["pins"] for ent in ft._ENGINES.entries)
    ft._ENGINES.clear()
    assert e0._raw_stream is None                                   # the raw HIP priority stream was destroyed on eviction


# ------------------------------------------------------------------------------------------------ the reference-shaped loop

class _Loader:
    def __init__(self, episodes):
        self.episodes = episodes

    def __len__(self):
        return len(self.episodes)

    def __iter__(self):
        y = torch.zeros(5, 20)
        for ep in self.episodes:
            yield [(v, y) for v in ep]


@pytest.mark.parametrize("method", ["gnnnet", "all"])
def test_reference_shaped_loop_with_lookahead_equals_batched(method):
    """finetune.py:599-632 / :634-666 verbatim in shape -- one finetune() (and finetune_linear()) call per episode, result read
    at once -- over a LookaheadLoader: bit-identical to scores_batched on the same episodes and numpy stream."""
    ft.params = argparse.Namespace(model='ResNet10', fine_tune_epoch=1, method=method)
    sd = synthetic.gnnnet_state_dict(seed=13)
    sd_b = synthetic.gnnnet_state_dict(seed=403)
    model = _model(sd)
    eps = [synthetic.test_episode(700 + i, 5, 5, 15, 84, gen_examples=1) for i in range(5)]
    torch.manual_seed(5)
    cls = ft.classifier_init(5, n=5)
    np.random.seed(10)
    ref = ft.scores_batched(method, [[v.cuda() for v in ep] for ep in eps], model, sd, sd_b, 1, 5, 5, 2, classifiers=cls)
    np.random.seed(10)
    novel_loader = ft.LookaheadLoader(_Loader(eps), method, model, sd, sd_b, 1, 5, 5, episodes_per_batch=2, classifiers=cls)
    acc_all, got = [], []
    for idx, elem in enumerate(novel_loader):                                   # finetune.py:599
        assert torch.all(torch.eq(elem[0][0], elem[1][0]))                      # :606
        _, y = elem[0]
        liz_x = [x for (x, y) in elem]
        if method == "all":                                                      # :647-649
            scores = ft.finetune_linear(liz_x, y, state_in=sd_b, linear=True, save_it=-1, n_query=15, n_way=5, n_support=5)
            scores = scores + ft.finetune(liz_x, y, model, sd, save_it=600, n_query=15, n_way=5, n_support=5)
        else:                                                                    # :619
            scores = ft.finetune(liz_x, y, model, sd, save_it=-1, n_query=15, n_way=5, n_support=5)
        topk_scores, topk_labels = scores.data.topk(1, 1, True, True)            # :625-628
        topk_ind = topk_labels.cpu().numpy()
        acc_all.append(float(np.sum(topk_ind[:, 0] == np.repeat(range(5), 15))) / 75 * 100)
        got.append(scores.clone())
    assert len(got) == 5 and not ft._READY
    assert torch.equal(torch.stack(got), ref)
    assert model.n_query == 15


# ------------------------------------------------------------------------------------------------ 20-shot / 50-shot configs

@pytest.mark.parametrize("E_epochs,G_aug", [(0, 0), (1, 0), (1, 1)])
def test_engine_20shot_vs_reference_golden(golden_dir, E_epochs, G_aug):
    """BASELINE configs[2] through FinetuneEngine (N=105 graph, 60-80 inner steps) against the reference's finetune()."""
    g = _g(golden_dir, "g13_finetune_20shot.npz")
    sd = synthetic.gnnnet_state_dict(seed=113)
    liz = synthetic.test_episode(141 + G_aug, 5, 20, 15, 84, gen_examples=G_aug)
    e = eng.FinetuneEngine(sd, n_support=20, n_views=2 + G_aug, fine_tune_epoch=E_epochs, episodes_per_batch=2)
    np.random.seed(10)
    sc = e.run_batch([liz])[0].cpu().numpy()
    ref = g["scores_E%d_G%d" % (E_epochs, G_aug)]
    if E_epochs == 0:
        np.testing.assert_allclose(sc, ref, atol=1e-4)
    else:
        err = np.abs(sc - ref)
        assert err.max() < 2e-2 and (sc.argmax(1) == ref.argmax(1)).mean() >= 0.96, err.max()
    e.close()


@pytest.mark.parametrize("E_epochs", [0, 1])
def test_finetune_50_vs_reference_golden(golden_dir, E_epochs):
    """BASELINE configs[4]: finetune_50.finetune() with gnnnet_copy.GnnNet (true n_support 50 -> folded N=130 graph) against
    the reference's finetune_50.finetune(); and the float64 envelope for the adapted case."""
    g = _g(golden_dir, "g14_finetune_50shot.npz")
    sd = synthetic.gnnnet_state_dict(seed=213)
    model = _model(sd, 50, gnnnet_copy.GnnNet)
    assert model.n_support == 25 and model.FOLD50
    liz = synthetic.test_episode(241, 5, 50, 15, 84, gen_examples=0)
    ft50.params = argparse.Namespace(model='ResNet10', fine_tune_epoch=E_epochs)
    np.random.seed(10)
    st = np.random.get_state()
    sc = ft50.finetune(liz, None, model, sd, None, n_query=15, n_way=5, n_support=50).cpu().numpy()
    ref = g["scores_E%d_G0" % E_epochs]
    if E_epochs == 0:
        np.testing.assert_allclose(sc, ref, atol=1e-4)
    else:
        err = np.abs(sc - ref)
        assert err.max() < 2e-2 and (sc.argmax(1) == ref.argmax(1)).mean() >= 0.96, err.max()
        np.random.set_state(st)
        o64 = O.finetune_episode(sd, liz, 5, 50, total_epoch=1, dtype=torch.float64, fold50=True).numpy()
        assert np.abs(sc - o64).max() <= max(4.0 * np.abs(ref - o64).max(), 2e-3)
    ft._ENGINES.clear()


def test_engine_50shot_batched_two_episodes():
    """FinetuneEngine(n_support=50, fold50=True) with two episodes in lockstep equals two single-episode runs (per-episode
    BatchNorm groups, per-episode GNN statistics)."""
    sd = synthetic.gnnnet_state_dict(seed=213)
    eps = [synthetic.test_episode(260 + i, 5, 50, 15, 84, gen_examples=0) for i in range(2)]
    rs = np.random.RandomState(3)
    perms = [[rs.permutation(750)] for _ in range(2)]              # 5 x 50 supports x (2 views + 1)
    e2 = eng.FinetuneEngine(sd, n_support=50, n_views=2, fine_tune_epoch=1, episodes_per_batch=2, fold50=True)
    both = e2.run_batch(eps, perms=perms).cpu().numpy()
    e2.close()
    e1 = eng.FinetuneEngine(sd, n_support=50, n_views=2, fine_tune_epoch=1, episodes_per_batch=1, fold50=True)
    for i in range(2):
        one = e1.run_batch([eps[i]], perms=[perms[i]])[0].cpu().numpy()
        assert np.abs(one - both[i]).max() < 2e-3 and (one.argmax(1) == both[i].argmax(1)).mean() >= 0.97
    e1.close()


# ------------------------------------------------------------------------------------------------ CLI surface

@pytest.mark.parametrize("argv", [
    ["--method", "gnnnet", "--n_shot", "5"],
    ["--method", "baseline", "--n_shot", "5"],
    ["--method", "all", "--n_shot", "5"],
    ["--method", "all", "--n_shot", "5", "--freeze_backbone"],
    ["--method", "gnnnet", "--n_shot", "50"],
])
def test_finetune_main_cli(argv, monkeypatch, capsys):
    """finetune.py:424-682's command line: every README method value, --freeze_backbone, --n_shot 50 (gnnnet_copy)."""
    monkeypatch.setenv("MFT_EPISODES", "3")
    monkeypatch.setenv("MFT_EPISODES_PER_BATCH", "2")
    accs = ft.main(argv + ["--fine_tune_epoch", "1", "--gen_examples", "1", "--model", "ResNet10"])
    out = capsys.readouterr().out
    assert len(accs) == 3 and np.all((accs >= 0) & (accs <= 100))
    assert "3 Test Acc = " in out and out.splitlines()[0] in ("True", "False")
    ft._ENGINES.clear(); ft._LIN_ENGINES.clear()


def test_finetune_50_main_cli(monkeypatch, capsys):
    monkeypatch.setenv("MFT_EPISODES", "2")
    monkeypatch.setenv("MFT_EPISODES_PER_BATCH", "2")
    accs = ft50.main(["--method", "gnnnet", "--n_shot", "50", "--fine_tune_epoch", "1", "--gen_examples", "0", "--model", "ResNet10"])
    assert len(accs) == 2 and ft50.params.n_shot == 50
    ft._ENGINES.clear()


def test_unknown_method_is_refused():
    with pytest.raises(NotImplementedError):
        ft.main(["--method", "protonet"])


# ------------------------------------------------------------------------------------------------ multi-GPU wiring

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, out, nproc):
    env = dict(os.environ, MFT_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    if nproc == 1:
        cmd = [sys.executable, worker, "--mode", mode, "--out", out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), worker, "--mode", mode, "--out", out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_rank_sharded_evaluation_equals_one_rank(tmp_path):
    """finetune.evaluate under torchrun with 2 ranks (episode i -> rank i mod 2, per-episode permutation / classifier streams,
    one all-gather of accuracies) returns on every rank exactly the accuracies of the 1-rank run -- on real engine outputs
    (--method all: both engines), fresh child processes."""
    _run_ranks("finetune", str(tmp_path / "w2"), 2)
    _run_ranks("finetune", str(tmp_path / "w1"), 1)
    a0 = np.load(str(tmp_path / "w2.0.npz"))["accs"]
    a1 = np.load(str(tmp_path / "w2.1.npz"))["accs"]
    b = np.load(str(tmp_path / "w1.0.npz"))["accs"]
    assert a0.shape == (6,) and np.array_equal(a0, a1)
    assert np.array_equal(a0, b), (a0, b)


def test_two_rank_meta_training_equals_accumulate_emulation(tmp_path):
    """train.main under torchrun with 2 ranks: each rank runs its own episode, one flat-bucket all-reduce of the outer
    gradients / 2, the same fused Adam step on both ranks.  Both ranks end with identical parameters, and they equal the
    single-process emulation "accumulate the 2 episodes' gradients from common parameters, divide by 2, one Adam step"."""
    _run_ranks("train", str(tmp_path / "t2"), 2)
    p0 = np.load(str(tmp_path / "t2.0.npz"))
    p1 = np.load(str(tmp_path / "t2.1.npz"))
    for k in p0.files:
        assert np.array_equal(p0[k], p1[k]), k
    # emulation in this process: 2 epoch-steps x 2 episodes each (n_episode = 4: rank r takes episodes r, r+2)
    from meta_fine_tuning_amd import optim
    torch.manual_seed(0)
    model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda()
    model.train()
    opt = optim.Adam(model.parameters())
    for step in range(2):
        grads = None
        for r in range(2):
            x = synthetic.train_episode(r + 2 * step, 5, 5, 16, 84)
            model.n_query = 16
            opt.zero_grad()
            model.set_forward_loss(x).backward()
            g = [p.grad.clone() for p in model.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(model.parameters(), grads):
            p.grad = g / 2
        opt.step()
    named = dict(model.named_parameters())
    for k in p0.files:
        d = np.abs(named[k].detach().cpu().numpy() - p0[k])
        # two Adam steps of <= lr = 1e-3 each; identical up to summation-order rounding on near-zero gradients
        assert (d < 2e-5).mean() > 0.995 and d.max() <= 2.1e-3, (k, d.max(), (d < 2e-5).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["finetune", "metatrain"])
def test_bench_two_ranks_one_json_line(workload):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU; here both ranks on one device over
    gloo, the MFT_BENCH_ONE_DEVICE hook): barrier + device synchronisation around the timed steps, MAX over ranks, rank 0 prints
    ONE JSON line whose value is the whole-job rate over both ranks."""
    import json
    env = dict(os.environ, MFT_BENCH_ONE_DEVICE="1", MFT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", workload]
    if workload == "finetune":
        cmd += ["--episodes-per-batch", "8", "--epochs", "1", "--gen-examples", "2", "--no-standalone", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "episodes/s" and d["value"] > 0 and d["vs_baseline"] is None
    per_step = 2 * (8 if workload == "finetune" else 1)                  # episodes of BOTH ranks per step
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) <= 0.02 * d["value"]
    if workload == "finetune":
        assert d["config"]["episodes_total"] == 2 * 8 * 2 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
        assert "cpu_baseline" not in d                                   # rank 0 at N = 1 only
