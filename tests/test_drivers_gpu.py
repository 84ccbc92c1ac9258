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
    monkeypatch.setenv("MFT_STANDIN_WEIGHTS", "1")                    # no checkpoints on this box: explicit opt-in
    accs = ft.main(argv + ["--fine_tune_epoch", "1", "--gen_examples", "1", "--model", "ResNet10"])
    out = capsys.readouterr().out
    assert len(accs) == 3 and np.all((accs >= 0) & (accs <= 100))
    assert "3 Test Acc = " in out and out.splitlines()[0] in ("True", "False")
    assert "SYNTHETIC stand-in weights" in [ln for ln in out.splitlines() if "Test Acc" in ln][0]     # never an untagged number
    ft._ENGINES.clear(); ft._LIN_ENGINES.clear()


def test_finetune_50_main_cli(monkeypatch, capsys):
    monkeypatch.setenv("MFT_EPISODES", "2")
    monkeypatch.setenv("MFT_EPISODES_PER_BATCH", "2")
    monkeypatch.setenv("MFT_STANDIN_WEIGHTS", "1")
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


def _run_ranks(mode, out, nproc, **extra_env):
    env = dict(os.environ, MFT_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **extra_env)
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


def test_rccl_collectives_execute_on_one_rank(tmp_path):
    """backend "nccl" (RCCL) on this box's single GPU, collectives FORCED at world size 1 (MFT_FORCE_COLLECTIVES=1): communicator
    set-up, the flat 21 MB-style gradient bucket all-reduce, the float64 accuracy all-gather on the device, parameter / buffer
    broadcasts, then finetune.evaluate and a train.main epoch on that process group.  Every multi-GPU code path of the drivers
    has then executed against the real library (the 2-rank tests above run gloo on one device); values equal the plain
    single-process run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", MFT_FORCE_COLLECTIVES="1", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("MFT_ONE_DEVICE", None)
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    r = subprocess.run([sys.executable, worker, "--mode", "rccl1", "--out", str(tmp_path / "r1")], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(str(tmp_path / "r1.0.npz"))
    assert int(z["ok_allreduce"]) == 1 and int(z["ok_broadcast"]) == 1
    assert np.array_equal(z["vals"], [1.5, 2.5, 99.0])
    # the same evaluation / training epoch without a process group
    env2 = {k: v for k, v in env.items() if k not in ("MFT_FORCE_COLLECTIVES", "RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r2 = subprocess.run([sys.executable, "-c",
                         "import sys, numpy as np, torch, tempfile; sys.path.insert(0, %r); import meta_fine_tuning_amd\n"
                         "from meta_fine_tuning_amd import finetune, synthetic, train, configs\n"
                         "from meta_fine_tuning_amd.io_utils import model_dict\n"
                         "from meta_fine_tuning_amd.methods.gnnnet import GnnNet\n"
                         "state = synthetic.gnnnet_state_dict(seed=0)\n"
                         "model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda(); model.load_state_dict(state)\n"
                         "accs = finetune.evaluate(model, state, 3, 5, 5, 15, 84, 1, 1, seed0=500, episodes_per_batch=2, verbose=False, method='gnnnet', rng_seed=10)\n"
                         "configs.save_dir = tempfile.mkdtemp(); torch.manual_seed(0)\n"
                         "m = train.main(['--method', 'gnnnet', '--model', 'ResNet10', '--stop_epoch', '1', '--save_freq', '1'], n_episode=2, size=84)\n"
                         "np.savez(%r, accs=accs, fc=m.state_dict()['fc.0.weight'].detach().cpu().numpy())\n" % (ROOT, str(tmp_path / "plain.npz"))],
                        env=env2, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    p = np.load(str(tmp_path / "plain.npz"))
    assert np.array_equal(z["accs"], p["accs"]) and np.array_equal(z["fc"], p["fc"])


def test_two_rank_meta_training_equals_accumulate_emulation(tmp_path):
    """train.main under torchrun with 2 ranks: each rank runs its own episode, one flat-bucket all-reduce of the outer
    gradients / 2, the same fused Adam step on both ranks.  Both ranks end with identical parameters, and they equal the
    single-process emulation "accumulate the 2 episodes' gradients from common parameters, divide by 2, one Adam step"."""
    _run_ranks("train", str(tmp_path / "t2"), 2)
    p0 = dict(np.load(str(tmp_path / "t2.0.npz")))
    p1 = dict(np.load(str(tmp_path / "t2.1.npz")))
    assert int(p0.pop("graphed")) == 0 and int(p1.pop("graphed")) == 0          # two steps per rank: still in the eager warm-up
    for k in p0:
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
    for k in p0:
        d = np.abs(named[k].detach().cpu().numpy() - p0[k])
        # two Adam steps of <= lr = 1e-3 each; identical up to summation-order rounding on near-zero gradients
        assert (d < 2e-5).mean() > 0.995 and d.max() <= 2.1e-3, (k, d.max(), (d < 2e-5).mean())


def test_two_rank_lockstep_meta_training_equals_accumulate_emulation(tmp_path):
    """train.main --episodes_per_rank 2 under torchrun with 2 ranks (round 6): every step each rank runs TWO episodes in lockstep,
    the flat bucket sums the ranks, 1 / W is applied inside the fused Adam launch -- the step of FOUR episodes.  Both ranks end
    identical, and equal the single-process emulation that accumulates the four episodes' gradients from common parameters."""
    _run_ranks("train", str(tmp_path / "l2"), 2, MFT_TEST_EPISODES_PER_RANK="2")
    p0 = dict(np.load(str(tmp_path / "l2.0.npz")))
    p1 = dict(np.load(str(tmp_path / "l2.1.npz")))
    p0.pop("graphed"); p1.pop("graphed")
    for k in p0:
        assert np.array_equal(p0[k], p1[k]), k
    from meta_fine_tuning_amd import optim
    torch.manual_seed(0)
    model = GnnNet(model_dict['ResNet10'], n_way=5, n_support=5).cuda()
    model.train()
    opt = optim.Adam(model.parameters())
    # the ranks' streams: rank r takes episodes r, r + 2, r + 4, ...; its LockstepLoader groups two consecutive ones per step.  The
    # emulation runs each rank's pair through the SAME lockstep launches (what the lockstep launches compute is pinned against
    # single-episode runs by tests/test_metatrain_gpu.py::test_lockstep_*; after Adam's first steps -- lr * sign(g) -- a parameter
    # comparison across DIFFERENT fp32 evaluations of a near-zero gradient would only measure sign flips)
    for step in range(2):
        grads = None
        for r in range(2):
            xs = torch.stack([synthetic.train_episode(r + 2 * (2 * step + j), 5, 5, 16, 84) for j in range(2)])
            model.n_query = 16
            opt.zero_grad()
            model.set_forward_loss_lockstep(xs).backward()
            g = [p.grad.clone() for p in model.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(model.parameters(), grads):
            p.grad = g / 2
        opt.step()
    named = dict(model.named_parameters())
    for k in p0:
        d = np.abs(named[k].detach().cpu().numpy() - p0[k])
        assert (d < 2e-5).mean() > 0.995 and d.max() <= 2.1e-3, (k, d.max(), (d < 2e-5).mean())


def test_two_rank_meta_training_graphed_equals_eager(tmp_path):
    """Episode-parallel meta-training with the forward + backward replayed from a hipGraph (captured on every rank's 4th step;
    the flat-bucket all-reduce and the fused Adam step stay outside the graph) against the same 2-rank run with eager launches:
    six steps per rank, both ranks, bit for bit."""
    _run_ranks("train", str(tmp_path / "g"), 2, MFT_TEST_TRAIN_STEPS="6", MFT_TRAIN_GRAPH="1")
    _run_ranks("train", str(tmp_path / "e"), 2, MFT_TEST_TRAIN_STEPS="6", MFT_TRAIN_GRAPH="0")
    for r in range(2):
        g, e = np.load(str(tmp_path / ("g.%d.npz" % r))), np.load(str(tmp_path / ("e.%d.npz" % r)))
        assert int(g["graphed"]) == 1 and int(e["graphed"]) == 0
        for k in e.files:
            if k != "graphed":
                assert np.array_equal(g[k], e[k]), (r, k)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,launcher", [("finetune", "self"), ("finetune", "torchrun"), ("metatrain", "self")])
def test_bench_two_ranks_one_json_line(workload, launcher):
    """bench.py for N > 1, both ways in: ``launcher == "torchrun"`` is how the driver starts it (torch.distributed.run, one rank
    per GPU); ``"self"`` is the PLAIN command `python bench.py --gpus 2` -- a parent that never touches the GPU spawns the ranks,
    relays their output and exits with their status.  Here both ranks share one device over gloo (MFT_BENCH_ONE_DEVICE hook).
    Barrier + device synchronisation around the timed steps, MAX over ranks, rank 0 prints ONE JSON line whose value is the
    whole-job rate over both ranks."""
    import json
    env = dict(os.environ, MFT_BENCH_ONE_DEVICE="1", MFT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
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


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["finetune", "metatrain"])
def test_bench_collectives_on_rccl_with_one_rank(workload):
    """bench.py's N-rank code path on backend "nccl" (RCCL) with ONE rank (MFT_FORCE_COLLECTIVES=1): process group bound to the
    device, barriers around the timed steps, the MAX all-reduce of the time, the all-gather of accuracies (finetune) / the flat
    gradient-bucket all-reduce in front of every outer step (metatrain) -- executed against the real library on this box's GPU,
    which the gloo one-device tests above cannot do."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", MFT_FORCE_COLLECTIVES="1", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("MFT_BENCH_ONE_DEVICE", "MFT_DIST_BACKEND", "MFT_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", workload]
    if workload == "finetune":
        cmd += ["--episodes-per-batch", "8", "--epochs", "1", "--gen-examples", "2", "--no-standalone", "--no-cpu-baseline", "--strong-episodes", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "episodes/s"


def test_bench_self_launch_propagates_failure():
    """The self-launching parent exits with the ranks' status: without the one-device hook rank 1 of `--gpus 2` asks for cuda:1,
    which a one-GPU box does not have (on a multi-GPU node the command simply succeeds and the test has nothing to show)."""
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a box with exactly one GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "MFT_BENCH_ONE_DEVICE", "MFT_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload",
                        "metatrain"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


# ------------------------------------------------------------------------------------------------ checkpoints, pipelined loop

def test_train_main_then_finetune_main_round_trip(tmp_path, monkeypatch, capsys):
    """What train.main writes is what finetune.main evaluates (finetune.py:448-527 lookup): two epochs of meta-training, then
    `finetune.main --save_iter 1` reads <save_dir>/checkpoints/miniImageNet/ResNet10_gnnnet_5way_5shot/1.tar and returns exactly
    the accuracies of loading that .tar by hand."""
    from meta_fine_tuning_amd import configs, train
    monkeypatch.setattr(configs, "save_dir", str(tmp_path))
    train.main(["--dataset", "miniImageNet", "--method", "gnnnet", "--model", "ResNet10", "--stop_epoch", "2", "--save_freq", "1"],
               n_episode=2, size=84)
    f = tmp_path / "checkpoints" / "miniImageNet" / "ResNet10_gnnnet_5way_5shot" / "1.tar"
    assert f.is_file() and (f.parent / "0.tar").is_file()
    capsys.readouterr()                                               # (train's "Epoch ... Loss" lines)
    monkeypatch.setenv("MFT_EPISODES", "3")
    monkeypatch.setenv("MFT_EPISODES_PER_BATCH", "2")
    ft._ENGINES.clear()
    accs = ft.main(["--method", "gnnnet", "--save_iter", "1", "--fine_tune_epoch", "1", "--gen_examples", "1", "--model", "ResNet10"])
    cap = capsys.readouterr()
    assert ft.main.loaded["gnnnet"] == str(f) and ("loading gnnnet checkpoint %s" % f) in cap.err
    assert cap.out.splitlines()[0] == "False" and "3 Test Acc = " in cap.out            # stdout: the reference's lines only
    assert "SYNTHETIC" not in cap.out
    state = torch.load(str(f), map_location="cpu")["state"]
    state = {k: v for k, v in state.items() if "feature2." not in k and "feature3." not in k}
    model = _model(state)
    np.random.seed(10)
    ft._ENGINES.clear()
    ref = ft.evaluate(model, state, 3, 5, 5, 15, 84, 1, 1, episodes_per_batch=2, verbose=False, method="gnnnet",
                      device_episodes=True, balance=True)
    assert np.array_equal(accs, ref)
    # a mis-pointed save_dir is an error (the reference's torch.load raises), never a plausible number
    monkeypatch.setattr(configs, "save_dir", str(tmp_path / "empty"))
    monkeypatch.delenv("MFT_STANDIN_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError):
        ft.main(["--method", "gnnnet", "--save_iter", "1", "--fine_tune_epoch", "1", "--gen_examples", "1", "--model", "ResNet10"])
    # the trained weights are not the stand-ins: the (opted-in) stand-in run differs
    monkeypatch.setenv("MFT_STANDIN_WEIGHTS", "1")
    accs2 = ft.main(["--method", "gnnnet", "--save_iter", "1", "--fine_tune_epoch", "1", "--gen_examples", "1", "--model", "ResNet10"])
    assert ft.main.loaded["gnnnet"] is None and "synthetic stand-in weights" in capsys.readouterr().err
    assert accs2.mean() > 40.0                                        # G9's meta-trained head, not a random one (chance = 20 %)
    # an epoch that was never written, in a directory that exists: the reference's torch.load raises
    monkeypatch.setattr(configs, "save_dir", str(tmp_path))
    monkeypatch.delenv("MFT_STANDIN_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError):
        ft.main(["--method", "gnnnet", "--save_iter", "5", "--fine_tune_epoch", "1", "--gen_examples", "1"])
    ft._ENGINES.clear()


def test_pipelined_evaluate_equals_batch_by_batch():
    """finetune.evaluate's software pipeline (episodes generated two batches ahead on a side stream, next batch's ingest + stem
    cache prefetched, final passes deferred, one read-back at the end) returns exactly what the plain per-batch
    scores_batched calls return on the same episodes and permutation streams."""
    sd = synthetic.gnnnet_state_dict(seed=13)
    model = _model(sd)
    ft._ENGINES.clear()
    accs = ft.evaluate(model, sd, 7, 5, 5, 15, 84, 1, 1, seed0=900, episodes_per_batch=2, verbose=False, method="gnnnet",
                       rng_seed=10, device_episodes=True)
    from meta_fine_tuning_amd import parallel
    y_query = np.repeat(range(5), 15)
    ref = []
    for c in range(0, 7, 2):
        ids = list(range(c, min(c + 2, 7)))
        eps = [synthetic.test_episode_device(900 + i, "cuda", 5, 5, 15, 84, 1) for i in ids]
        sc = ft.scores_batched("gnnnet", eps, model, sd, None, 1, 5, 5, 2, rngs=[parallel.episode_rng(10, i) for i in ids])
        ref += [float(np.mean(p == y_query)) * 100 for p in sc.argmax(2).cpu().numpy()]
    assert np.array_equal(accs, np.asarray(ref))
    ft._ENGINES.clear()


def test_strong_scaling_leg_small():
    """bench.strong_scaling_leg on a tiny fixed job (one rank): the record carries the wall time of the whole evaluation."""
    import bench
    ft._ENGINES.clear()
    r = bench.strong_scaling_leg(4, bench.g9_state(), 0, 1, "cuda:0", e_max=2)
    assert r["scaling"] == "strong" and r["episodes"] == 4 and r["batches_per_rank"] == 2 and r["episodes_per_batch"] == 2
    assert r["wall_s"] > 0 and abs(r["episodes_per_s"] - 4 / r["wall_s"]) < 0.05 and r["mean_acc"] > 40.0


def test_lookahead_loader_with_freeze_backbone_runs_the_frozen_path():
    """The reference's loop body passes freeze_backbone on every call (finetune.py:615-619): a LookaheadLoader built with it
    yields the loader untouched and the per-episode calls score with eval-mode features, exactly as without the wrapper."""
    ft.params = argparse.Namespace(model='ResNet10', fine_tune_epoch=1, method="gnnnet")
    sd = synthetic.gnnnet_state_dict_with_running_stats(seed=13)
    model = _model(sd)
    eps = [synthetic.test_episode(720 + i, 5, 5, 15, 84, gen_examples=1) for i in range(2)]
    np.random.seed(10)
    ref = [ft.finetune(ep, None, model, sd, None, freeze_backbone=True) for ep in eps]
    np.random.seed(10)
    got = []
    for elem in ft.LookaheadLoader(_Loader(eps), "gnnnet", model, sd, None, 1, 5, 5, episodes_per_batch=2, freeze_backbone=True):
        liz_x = [x for (x, y) in elem]
        got.append(ft.finetune(liz_x, None, model, sd, None, freeze_backbone=True))
    assert all(torch.equal(a, b) for a, b in zip(got, ref)) and not ft._READY
    # and the non-frozen wrapper refuses a frozen call instead of handing out fine-tuned scores
    it = iter(ft.LookaheadLoader(_Loader(eps), "gnnnet", model, sd, None, 1, 5, 5, episodes_per_batch=2))
    elem = next(it)
    with pytest.raises(RuntimeError):
        ft.finetune([x for (x, y) in elem], None, model, sd, None, freeze_backbone=True)
    it.close()
    ft._READY.clear()
    ft._ENGINES.clear()


# ------------------------------------------------------------------------------------------------ dataset-shaped episode sources

def test_train_main_samples_episodes_from_the_resident_pool(tmp_path, monkeypatch, capsys):
    """`train.py --dataset miniImageNet [--train_aug]` (BASELINE configs[3]): episodes come from the miniImageNet-shaped uint8
    pool resident in HBM (train.ResidentEpisodeLoader: randperm(64)[:5] classes, 21 distinct images per class, training-side
    transform on the device), --train_aug switches the transform AND the checkpoint directory (train.py:176-177)."""
    from meta_fine_tuning_amd import configs, train
    monkeypatch.setattr(configs, "save_dir", str(tmp_path))
    seen = []
    orig = train.ResidentEpisodeLoader.episode

    def spy(self, epoch, i):
        x = orig(self, epoch, i)
        seen.append((self.aug, tuple(x.shape), x.is_cuda, float(x.float().std())))
        return x
    monkeypatch.setattr(train.ResidentEpisodeLoader, "episode", spy)
    for aug in (False, True):
        seen.clear()
        args = ["--dataset", "miniImageNet", "--method", "gnnnet", "--model", "ResNet10", "--stop_epoch", "1", "--save_freq", "1"]
        m = train.main(args + (["--train_aug"] if aug else []), n_episode=3, size=84, pool_images_per_class=25)
        d = tmp_path / "checkpoints" / "miniImageNet" / ("ResNet10_gnnnet_aug_5way_5shot" if aug else "ResNet10_gnnnet_5way_5shot")
        assert (d / "0.tar").is_file()
        assert len(seen) == 3 and all(s[0] is aug and s[1] == (5, 21, 3, 84, 84) and s[2] and s[3] > 0.1 for s in seen)
        out = capsys.readouterr().out
        assert "Loss" in out and "nan" not in out.lower()
        assert all(torch.isfinite(p).all() for p in m.parameters())


def test_train_main_episodes_per_rank_lockstep(tmp_path, monkeypatch, capsys):
    """`train.py --episodes_per_rank 2` (round 6, not in the reference): six episodes of the resident pool become three optimizer
    steps of two episodes in lockstep; the checkpoint is written as usual, the loss lines count steps, parameters stay finite and
    move, and the run equals the same driver with the hipGraph replay off (eager lockstep steps)."""
    from meta_fine_tuning_amd import configs, graph_step, train
    outs = []
    for graphed in (True, False):
        monkeypatch.setattr(graph_step, "ENABLED", graphed)
        monkeypatch.setattr(configs, "save_dir", str(tmp_path / ("g%d" % graphed)))
        np.random.seed(10)
        torch.manual_seed(0)
        m = train.main(["--dataset", "miniImageNet", "--method", "gnnnet", "--model", "ResNet10", "--stop_epoch", "1", "--save_freq", "1",
                        "--episodes_per_rank", "2"], n_episode=10, size=84, pool_images_per_class=25)
        d = tmp_path / ("g%d" % graphed) / "checkpoints" / "miniImageNet" / "ResNet10_gnnnet_5way_5shot"
        assert (d / "0.tar").is_file()
        out = capsys.readouterr().out
        assert "Epoch 0 | Batch 0/5 | Loss" in out and "nan" not in out.lower()
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, out))
        assert all(torch.isfinite(p).all() for p in m.parameters())
    (a, pa), (b, pb) = outs
    assert pa == pb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_finetune_main_test_dataset_uses_the_resident_sampler(monkeypatch, capsys):
    """`finetune.py --test_dataset EuroSAT` (finetune.py:558-579; BASELINE configs[4] is EuroSAT-shaped): episodes are sampled
    from a EuroSAT-shaped uint8 pool in HBM (10 classes, 64x64) and their 2 + G views are generated on the device by the engine
    (run_batch(sources=True)); equal to calling evaluate() with the same sampler; an unknown name raises."""
    from meta_fine_tuning_amd import augment
    monkeypatch.setenv("MFT_STANDIN_WEIGHTS", "1")
    monkeypatch.setenv("MFT_EPISODES", "3")
    monkeypatch.setenv("MFT_EPISODES_PER_BATCH", "2")
    monkeypatch.setenv("MFT_POOL_PER_CLASS", "40")
    ft._ENGINES.clear()
    args = ["--method", "gnnnet", "--fine_tune_epoch", "1", "--gen_examples", "2", "--model", "ResNet10", "--test_dataset", "EuroSAT"]
    accs = ft.main(args)
    out = capsys.readouterr().out
    assert "Loading EuroSAT" in out and "3 Test Acc = " in out
    assert accs.shape == (3,) and np.all((accs >= 0) & (accs <= 100))
    pool = synthetic.class_pool_u8("EuroSAT", "cuda:0", seed=1, n_per_class=40)
    assert pool.shape == (10, 40, 64, 64, 3)
    sampler = augment.EpisodeSampler(pool, 5, 20, seed=10)
    state = ft.standin_state("gnn", 5)
    model = _model(state)
    np.random.seed(10)
    ft._ENGINES.clear()
    ref = ft.evaluate(model, state, 3, 5, 5, 15, 84, 2, 1, episodes_per_batch=2, verbose=False, method="gnnnet", balance=True,
                      sampler=sampler)
    assert np.array_equal(accs, ref)
    # the per-episode entry points (--method all / baseline, frozen backbone) get the same episodes as materialised views
    np.random.seed(10)
    fr = ft.evaluate(model, state, 2, 5, 5, 15, 84, 2, 1, episodes_per_batch=2, verbose=False, method="gnnnet", freeze_backbone=True,
                     sampler=sampler)
    assert fr.shape == (2,) and np.all((fr >= 0) & (fr <= 100))
    with pytest.raises(ValueError):
        ft.main(args[:-1] + ["Omniglot"])
    ft._ENGINES.clear()
