#!/usr/bin/env python3
"""bench.py -- episodes/sec of the test-time meta-fine-tuning hot path on MI355X.

Workload (BASELINE.json configs[1]): 5-way 5-shot ResNet10+GNN, fine_tune_epoch=5, gen_examples=17
(=> 500 Adam inner steps of 5 images per episode, finetune.py:270-299), 15 queries, 84x84 synthetic
episodes, followed by the 100-image transductive feature pass and the GNN head (finetune.py:306-317).
A "step" = one lockstep batch of E episodes through FinetuneEngine.run_batch.  Inputs (the E resident
synthetic episodes, NCHW fp32, all 19 views) are in HBM before the timed region; every step draws
fresh numpy permutations.  One process per GPU; episodes shard across ranks with no data-path
collective (only a final all-gather of per-episode accuracies), so scaling is "weak".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--episodes-per-batch E]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic FLOPs (2*MAC; conv + linear + per-pair MLP only) -- SURVEY.md §8(d)
F_IMG_84 = 0.28585e9          # ResNet10 forward per 84x84 image
F_LB_84 = 0.1086e9            # last-block backward per image (wgrad C1,C2,shortcut + dgrad C2)
F_GNN_15_30 = 8.08e9          # GNN forward, 15 graphs of 30 nodes
PEAK_F32_MFMA = 157.3e12      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA = 2500e12      # same guide: dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0         # same guide: HBM3E ~8 TB/s peak (spec); ~6.3 TB/s achievable read-only


def gnn_flops(B, N):
    """F_gnn(B,N) ~ 596,160*B*N^2 + 64,868*B*N (+ 1,086*B*N^2 for A@x)."""
    return (596160.0 + 1086.0) * B * N * N + 64868.0 * B * N


def episode_flops(n_way, n_shot, n_query, views, epochs):
    passes = epochs * n_way * n_shot * (views + 1)
    graph_support = n_shot // 2 if n_shot == 50 else n_shot          # gnnnet_copy.py:34 folds 50 -> 25 nodes per class
    f_gnn = F_GNN_15_30 if n_shot == 5 else gnn_flops(n_query, n_way * (graph_support + 1))
    return passes * (F_IMG_84 + F_LB_84) + n_way * (n_shot + n_query) * F_IMG_84 + f_gnn


def conv_flops(n_img, OH, Cout, K):
    return 2.0 * n_img * OH * OH * Cout * K


def kernel_source_sha(fused=False):
    """Identity of the dominant kernel's source (csrc/conv_igemm.hip -- or csrc/wgrad_fwd.hip for the opt-in fused form -- +
    csrc/mft_common.h): a PMC traffic figure is only quoted for byte-identical kernel code."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("meta-fine-tuning_amd/csrc/%s.hip" % ("wgrad_fwd" if fused else "conv_igemm"), "meta-fine-tuning_amd/csrc/mft_common.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(E, fused=False):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 corrections applied; profiles/pmc_traffic.json, written by tools/pmc_traffic_json.py together with
    the commit and the kernel-source hash it was measured on).  None unless a pass exists for this E AND for exactly this kernel
    source -- a number measured on other code is not reported."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            j = json.load(f)
        if int(j["episodes_per_step"]) == int(E) and j.get("kernel_source_sha16") == kernel_source_sha(fused):
            return {"mb_per_launch": j["traffic_mb_per_launch"], "algorithmic_mb_per_launch": j["algorithmic_mb_per_launch"],
                    "source": "profiles/pmc_traffic.json (round %s, measured at commit %s, kernel source %s)"
                              % (j["round"], j.get("head", "?"), j["kernel_source_sha16"])}
    except (OSError, KeyError, ValueError):
        pass
    return None


G9_NOISE, G9_SEED_SD, G9_EP_SEED0 = 2.0, 31, 90000          # oracle/make_golden_g9.py


# ---- algorithmic work of one launch, from the launcher's own C-ABI arguments (positions as declared in include/mft_hip.h;
# tests/test_host_cpu.py::test_bench_work_table_matches_the_header checks every index against the header's parameter names)
def _osz(h, k, s_, p_):
    return (h + 2 * p_ - k) // s_ + 1

def _adam_bytes(a, n=8, cin=11, cout=12, kh=13, kw=14, ipg=17):          # read w, m, v + write w, m, v; the gradient never reaches HBM
    return 24.0 * (a[n] // a[ipg]) * a[cout] * a[kh] * a[kw] * a[cin]

def _conv_fl(a, n, h, w_, cin, cout, kh, kw, st, pd):
    return 2.0 * a[n] * _osz(a[h], a[kh], a[st], a[pd]) * _osz(a[w_], a[kw], a[st], a[pd]) * a[cout] * a[kh] * a[kw] * a[cin]

LAUNCH_WORK = {   # launcher -> (family, algorithmic bytes | flops of one call from its C-ABI arguments)
    "mft_wgrad_adam_next_forward": ("adam", _adam_bytes),
    "mft_conv2d_wgrad_adam_nhwc": ("adam", _adam_bytes),
    "mft_conv2d_wgrad_adam_nhwc_dev": ("adam", _adam_bytes),
    "mft_conv2d_wgrad_adam_dgrad_nhwc": ("adam", lambda a: 24.0 * (a[8] // a[13]) * a[12] * 9 * a[11]),
    "mft_conv2d_wgrad_adam_dgrad_nhwc_dev": ("adam", lambda a: 24.0 * (a[8] // a[13]) * a[12] * 9 * a[11]),
    "mft_conv2d_nhwc": ("f32", lambda a: _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
    "mft_conv2d_nhwc_ksplit": ("f32", lambda a: _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
    "mft_conv2d_nhwc_ksplit_grouped": ("f32", lambda a: _conv_fl(a, 5, 6, 7, 8, 9, 10, 11, 12, 13)),
    "mft_conv2d_nhwc_x3": ("x3", lambda a: _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
    "mft_conv2d_nhwc_h2": ("x3", lambda a: _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
    "mft_conv2d_nhwc_x3_bnstats": ("x3", lambda a: _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
    "mft_conv2d_nhwc_h2_bnstats": ("x3", lambda a: _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)),
    # C2 of trunk.4 / trunk.5 with BN1 + ReLU applied by its loader (3x3, stride 1, pad 1): the same convolution FLOPs
    "mft_conv2d_nhwc_x3_bnin_bnstats": ("x3", lambda a: 2.0 * a[9] * a[10] * a[11] * a[13] * 9 * a[12]),
    "mft_conv2d_nhwc_h2_bnin_bnstats": ("x3", lambda a: 2.0 * a[9] * a[10] * a[11] * a[13] * 9 * a[12]),
}



def g9_state():
    """The weights of the accuracy golden G9 (seeded backbone + the head meta-trained with the REFERENCE's set_forward_loss,
    tests/golden/g9_head.npz): with them the engine's accuracies are comparable with the reference's own finetune() run."""
    from meta_fine_tuning_amd import synthetic
    sd = synthetic.gnnnet_state_dict(seed=G9_SEED_SD)
    hz = np.load(os.path.join(ROOT, "tests", "golden", "g9_head.npz"))
    for k in hz.files:
        sd[k] = torch.from_numpy(hz[k])
    return sd


def g9_episodes(n, dev, gen_examples=17, workers=None):
    """The first ``n`` episodes of golden G9 config B (numpy-generated, exactly the tensors the reference ran on), moved to HBM."""
    from concurrent.futures import ThreadPoolExecutor
    from meta_fine_tuning_amd import synthetic

    def make(i):          # (moved to the device by the worker that made it: the host never holds more than a few 160 MB episodes)
        return [v.to(dev) for v in synthetic.test_episode(G9_EP_SEED0 + i, 5, 5, 15, 84, gen_examples=gen_examples, noise=G9_NOISE)]
    with ThreadPoolExecutor(max_workers=workers or min(8, host_threads())) as ex:
        return list(ex.map(make, range(n)))


class PowerSampler:
    """Socket power and shader clock of the busiest GPU from its hwmon files (readable without privileges), sampled by a host
    thread over the timed region: the lockstep step runs AT the package power limit, so the clock the
    kernels get -- and with it the rate of the HBM-bound launches, ~10 B per cycle and CU -- depends on the lease's silicon and
    cooling.  Reported, not used."""

    def __init__(self):
        import glob
        self.cards = []
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            hw = glob.glob(card + "/hwmon/hwmon*")
            if hw and os.path.exists(hw[0] + "/power1_input"):
                self.cards.append((hw[0] + "/power1_input", hw[0] + "/freq1_input", hw[0] + "/power1_cap"))
        self.rows, self._stop, self._th = [], False, None

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return float("nan")

    def _run(self):
        while not self._stop:
            self.rows.append([(self._rd(pw) * 1e-6, self._rd(fq) * 1e-6) for pw, fq, _ in self.cards])
            time.sleep(0.05)

    def start(self):
        if self.cards:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join()
        if not self.rows:
            return None
        P = np.array([[c[0] for c in r] for r in self.rows])
        F = np.array([[c[1] for c in r] for r in self.rows])
        if not np.isfinite(P).any():
            return None
        ci = int(np.nanargmax(np.nanmean(P, axis=0)))
        cap = self._rd(self.cards[ci][2]) * 1e-6
        return {"socket_w_median": round(float(np.nanmedian(P[:, ci])), 0), "socket_w_max": round(float(np.nanmax(P[:, ci])), 0),
                "power_cap_w": None if not np.isfinite(cap) else round(cap, 0),
                "shader_mhz_median": round(float(np.nanmedian(F[:, ci])), 0), "shader_mhz_min": round(float(np.nanmin(F[:, ci])), 0),
                "samples": int(len(self.rows)),
                "what": "hwmon power1_input / freq1_input of the busiest GPU over the timed region (rank 0's host thread, 20 Hz)"}


def host_threads():
    """Threads the CPU baseline may use: the cores this process can actually run on (affinity mask and cgroup CPU
    quota), capped at 32 -- the 5-image convolutions of one inner step do not scale past that, and oversubscribing an
    OpenMP pool on a quota-limited box makes it pathologically slow."""
    from meta_fine_tuning_amd import parallel
    return parallel.host_threads(32)


def cpu_baseline_c1(state, cores, reps=3, budget_s=25.0):
    """C1 / C4 leg of SURVEY.md section 8(d): ONE meta-training step (train.py:28, meta_template.py:76-92: set_forward_loss +
    backward + Adam over all 104 tensors) on a 5-way 5-shot 16-query 84x84 episode, on the torch-CPU oracle with ``cores``
    threads; median of up to ``reps`` steps after one warm-up, bounded by ``budget_s``."""
    from oracle import mft_oracle as O
    from meta_fine_tuning_amd import synthetic
    torch.set_num_threads(cores)
    xe = synthetic.train_episode(5000, 5, 5, 16, 84)
    sd_t = O.clone_state(state)
    keys = [k for k, v in sd_t.items() if v.dtype.is_floating_point and "running_" not in k]
    ps = [sd_t[k].requires_grad_(True) for k in keys]
    ast = O.adam_init([p.detach() for p in ps])
    t_c1 = []
    t_all = time.perf_counter()
    for it in range(reps + 1):
        t0 = time.perf_counter()
        loss, _ = O.meta_train_loss(sd_t, xe, 5, 5)
        gs = torch.autograd.grad(loss, ps, allow_unused=True)
        with torch.no_grad():
            O.adam_step([p.detach() for p in ps], [g if g is not None else torch.zeros_like(p) for g, p in zip(gs, ps)], ast, lr=1e-3)
        if it:
            t_c1.append(time.perf_counter() - t0)
        if t_c1 and time.perf_counter() - t_all > budget_s:
            break
    return {"seconds": float(np.median(t_c1)), "steps_timed": len(t_c1), "threads": cores}


def cpu_baseline(state, episode, leg_budget_s=12.0):
    """the oracle (CPU restatement validated against the reference) timed on this box's host cores.
    C2 = ONE WHOLE episode of finetune() -- all 500 inner Adam steps, the 100-image pass, the GNN head -- at all granted cores and
    again at 8 threads (the core count of the container the reference's own numbers were taken in); C1 = one train_loop step
    (set_forward_loss + backward + Adam over all 104 tensors, median of 3) at all granted cores.  A leg that would exceed
    ``leg_budget_s`` stops after the steps it has done and says so (the rest is then extrapolated from its own step time), so a
    slow host cannot stall the GPU measurement."""
    from oracle import mft_oracle as O
    cores = host_threads()
    sd0 = O.clone_state(state)
    xa, ya = O.finetune_support_set(episode, 5, 5)
    n_total = xa.shape[0]
    x0 = episode[0]

    def whole_episode(threads):
        torch.set_num_threads(threads)
        sd_all = O.clone_state(sd0)
        fsd = O.feature_state(sd_all)
        adam = O.adam_init([fsd[k] for k in O.ADAPT_KEYS])
        rs = np.random.RandomState(0)
        sel0 = torch.from_numpy(rs.permutation(n_total)[:5])
        O.inner_step(O.feature_state(O.clone_state(sd0)), xa[sel0], ya[sel0], O.adam_init([fsd[k].clone() for k in O.ADAPT_KEYS]))   # warm
        t0 = time.perf_counter()
        done, n_steps = 0, 5 * ((n_total + 4) // 5)
        for ep in range(5):
            perm = rs.permutation(n_total)
            for j in range(0, n_total, 5):
                if time.perf_counter() - t0 > leg_budget_s and done >= 20:
                    break
                sel = torch.from_numpy(perm[j:j + 5])
                O.inner_step(fsd, xa[sel], ya[sel], adam)
                done += 1
        t_steps = time.perf_counter() - t0
        t1 = time.perf_counter()
        with torch.no_grad():
            feats = O.resnet10_forward(fsd, x0.reshape(100, *x0.shape[2:]), "", train=True).view(5, 20, -1)
            O.gnnnet_set_forward(sd_all, feats, 5, 5, 15, is_feature=True)
        t_final = time.perf_counter() - t1
        t_episode = t_steps * (n_steps / done) + t_final
        return {"threads": threads, "episodes_per_s": round(1.0 / t_episode, 5), "seconds_per_episode": round(t_episode, 2),
                "inner_steps_timed": done, "inner_steps_per_episode": n_steps, "ms_per_inner_step": round(t_steps / done * 1e3, 2),
                "final_pass_and_gnn_s": round(t_final, 2), "extrapolated": done < n_steps}

    legs = [whole_episode(cores)]
    if cores != 8:
        legs.append(whole_episode(min(8, cores) if cores < 8 else 8))
    c1 = cpu_baseline_c1(sd0, cores, reps=3)["seconds"]
    best = legs[0]
    return {"value": best["episodes_per_s"], "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": "ONE whole episode of BASELINE configs[1] (%d of %d inner Adam steps timed at %.1f ms each%s + the 100-image "
                      "pass + GNN head %.2f s = %.2f s per episode) on the torch-CPU oracle with %d threads (os.cpu_count()=%s)"
                      % (best["inner_steps_timed"], best["inner_steps_per_episode"], best["ms_per_inner_step"],
                         ", rest extrapolated" if best["extrapolated"] else "", best["final_pass_and_gnn_s"],
                         best["seconds_per_episode"], cores, os.cpu_count()),
            "legs": legs,
            "c1_train_loop_step": {"seconds": round(c1, 3), "episodes_per_s": round(1.0 / c1, 3), "threads": cores,
                                   "what": "one meta-training episode (105 images 84x84: set_forward_loss + backward + Adam over "
                                           "104 tensors), median of 3 after one warm-up"}}


def cpu_baseline_subprocess(gen_examples, timeout_s=240, workload="finetune"):
    """Run the CPU leg in a child process (own OpenMP pool, hard wall-clock limit) so that a slow host can never
    stall the GPU measurement; the child never initialises the GPU."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--gen-examples", str(gen_examples), "--workload", workload]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        note = "cpu baseline child failed rc=%d: %s" % (r.returncode, r.stderr[-300:])
    except subprocess.TimeoutExpired:
        note = "cpu baseline child exceeded %d s" % timeout_s
    return {"value": None, "unit": "episodes/s", "cores": host_threads(), "kind": "port", "sample": note}


def self_launch(n_gpus):
    """Start ``n_gpus`` ranks of this same command line under torch.distributed.run (127.0.0.1 rendezvous on a free port) from a
    parent that has not initialised any GPU; stdout / stderr of the ranks pass through; returns their exit status."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n_gpus)))
    return subprocess.call(cmd, env=env)


def strong_scaling_leg(n_episodes, state, rank, world, dev, e_max=128, emulate_world=0):
    """STRONG scaling next to the weak-scaling headline: a FIXED job -- the reference's 600-episode evaluation
    (finetune.py:593-682: `--method gnnnet --n_shot 5 --fine_tune_epoch 5 --gen_examples 17`) -- split over the ranks
    (episode i -> rank i mod W), timed from the call of finetune.evaluate to the gathered accuracies on every rank:
    engine build (slab placement included), on-device episode generation, all inner loops, final passes, the one all-gather.
    Episodes per lockstep batch = the rank's share split into equal batches of at most ``e_max``."""
    import torch.distributed as dist
    from meta_fine_tuning_amd import finetune as ft
    from meta_fine_tuning_amd.io_utils import model_dict
    from meta_fine_tuning_amd.methods.gnnnet import GnnNet
    model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=5).cuda()
    model.load_state_dict(state)
    ft._ENGINES.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    np.random.seed(10)
    tm = {}
    t0 = time.perf_counter()
    accs = ft.evaluate(model, state, n_episodes, 5, 5, 15, 84, 17, 5, seed0=7000, episodes_per_batch=e_max, verbose=False,
                       method="gnnnet", rng_seed=10, device_episodes=True, balance=True, timings=tm,
                       emulate_world=(0, emulate_world) if emulate_world else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if emulate_world:
        ft._ENGINES.clear()
        return {"what": "rank 0's share of the fixed %d-episode job at W = %d (episodes 0, %d, %d, ...: %d episodes) run ALONE on one GPU "
                        "through finetune.evaluate -- what each of the %d ranks does concurrently on its own GPU; the W-rank job's wall is "
                        "this plus the accuracy all-gather (600 doubles) and rank skew, which a 1-GPU box cannot measure; the ratio to the 1-GPU "
                        "leg can exceed W because a single-batch share skips the slab-placement scan (engine ready after 0.2 s instead of 0.7-1.2 s)"
                        % (n_episodes, emulate_world, emulate_world, 2 * emulate_world, len(accs), emulate_world),
                "emulated_world": emulate_world, "rank_share_episodes": int(len(accs)), "rank_wall_s": round(dt, 3),
                "projected_episodes_per_s": round(n_episodes / dt, 2), "episodes_per_batch": tm.get("episodes_per_batch"),
                "batches_per_rank": tm.get("batches"),
                "engine_ready_after_s": None if "engine_ready_s" not in tm else round(tm["engine_ready_s"], 3),
                "mean_acc_of_share": round(float(accs.mean()), 2)}
    if world > 1:
        t = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ft._ENGINES.clear()
    return {"what": "fixed job: %d episodes of BASELINE configs[1] over %d rank(s) through finetune.evaluate (engine build + slab "
                    "placement + on-device episode generation + inner loops + final passes + accuracy gather), MAX over ranks"
                    % (n_episodes, world),
            "scaling": "strong", "episodes": n_episodes, "n_gpus": world, "wall_s": round(dt, 3),
            "episodes_per_s": round(n_episodes / dt, 2), "episodes_per_batch": tm.get("episodes_per_batch"),
            "batches_per_rank": tm.get("batches"), "engine_ready_after_s": None if "engine_ready_s" not in tm else round(tm["engine_ready_s"], 3),
            "mean_acc": round(float(accs.mean()), 2)}


def strong_scaling_child(args, rank, world):
    """The fixed 600-episode job in a FRESH process per rank (what `python -m meta_fine_tuning_amd.finetune` under torchrun is),
    started before this process touches the GPU: a process that has already built and freed an engine hands the next one
    fragmented device memory (measured: the same job 13.5 s instead of 9-10 s when run after the weak-scaling part, and the
    weak-scaling part 60 instead of 79 episodes/s when run after the job).  The children form their own process group on
    MASTER_PORT + 1; rank 0's child prints the record, its parent adds the child's whole process wall time."""
    import subprocess
    env = dict(os.environ)
    if world > 1:
        env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
    cmd = [sys.executable, os.path.abspath(__file__), "--strong-only", "--gpus", str(world), "--strong-episodes", str(args.strong_episodes),
           "--episodes-per-batch", str(args.episodes_per_batch)]

    def child(extra):
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd + extra, env=env, capture_output=True, text=True, timeout=900)
        except subprocess.TimeoutExpired:
            return {"error": "strong-scaling child exceeded 900 s"}, 0.0
        wall = time.perf_counter() - t0
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "strong-scaling child failed rc=%d: %s" % (r.returncode, r.stderr[-300:])}, wall
        return json.loads(line[-1]), wall

    rec, wall = child([])
    if rank != 0:
        return None
    if "error" in rec:
        return rec
    rec["process_wall_s"] = round(wall, 3)
    rec["episodes_per_s_incl_process_start"] = round(rec["episodes"] / wall, 2)
    if world == 1 and args.emulate_world > 1:
        # the same job as ONE rank of a W-rank run would see it (fresh process again): projected W-GPU figure = 600 / that wall
        em, em_wall = child(["--emulate-world", str(args.emulate_world)])
        if "error" not in em:
            em["process_wall_s"] = round(em_wall, 3)
            em["projected_speedup_vs_this_1gpu_leg"] = round(rec["wall_s"] / em["rank_wall_s"], 2)
        rec["emulated_world_%d" % args.emulate_world] = em
    return rec


OTHER_CONFIGS = (
    # key, BASELINE.json config it measures, extra command line, timeout (s)
    ("configs2_20shot", "configs[2]: 5-way 20-shot ResNet10+GNN (N = 105 graph nodes), 2000 inner Adam steps per episode",
     ["--n-shot", "20", "--steps", "1", "--warmup", "1"], 420),
    ("configs4_50shot", "configs[4]: 5-way 50-shot compressed-GNN path (finetune_50.py: pair-averaged supports, N = 130), 5000 inner steps",
     ["--n-shot", "50", "--steps", "1", "--warmup", "1"], 420),
    ("configs3_metatrain", "configs[3]: meta-training step (set_forward_loss + full backward + flat-bucket all-reduce + fused outer Adam), "
     "one 105-image episode per rank per step, hipGraph replay", ["--workload", "metatrain", "--steps", "300", "--warmup", "10"], 300),
    ("configs3_metatrain_lockstep4", "configs[3] with the opt-in --episodes_per_rank 4: four episodes per optimizer step in lockstep on the one GPU "
     "(per-episode BatchNorm statistics, gradients averaged over them = the update of a 4-rank episode-parallel run; value counts 4 episodes "
     "per step)", ["--workload", "metatrain", "--episodes-per-rank", "4", "--steps", "200", "--warmup", "10"], 300),
    ("reference_224", "the reference's own image_size 224 (finetune.py:429; train.py:72) at configs[1]'s 5-way 5-shot, 500 inner steps",
     ["--image-size", "224", "--episodes-per-batch", "32", "--steps", "2", "--warmup", "1"], 420),
)


def other_configs_children(args):
    """BASELINE configs[2], [3], [4] and the reference's own 224x224 input, each as a SHORT run of this same script in a fresh child
    process (started before this process touches the GPU, like the fixed-job leg), so that the driver's one `python bench.py` line
    also carries driver-run numbers for them (round-5 verdict "missing 3"): value, ms per step, the kernel family that holds most
    of the step's launch time with its roofline fraction, and the rocprofv3 summary under profiles/ that backs the same command."""
    import subprocess
    out = {}
    for key, what, extra, tmo in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-standalone", "--strong-episodes", "0",
                                                                     "--validate-episodes", "0", "--no-other-configs"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=tmo)
        except subprocess.TimeoutExpired:
            out[key] = {"what": what, "error": "child exceeded %d s" % tmo}
            continue
        wall = time.perf_counter() - t0
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            out[key] = {"what": what, "error": "child failed rc=%d: %s" % (r.returncode, r.stderr[-300:])}
            continue
        j = json.loads(line[-1])
        rec = {"what": what, "command": "python bench.py " + " ".join(extra), "value": j["value"], "unit": j["unit"],
               "ms_per_step": j["ms_per_step"], "steps": j["steps"], "episodes_per_step": j["config"].get("episodes_per_step", 1),
               "process_wall_s": round(wall, 1)}
        roofs = {k: j[k] for k in ("roofline", "roofline_mfma", "roofline_mfma_x3") if j.get(k)}
        share = j.get("launch_time_share")
        dom = max(share, key=share.get) if share else "roofline"
        dom = {"adam": "roofline", "f32": "roofline_mfma", "x3": "roofline_mfma_x3"}.get(dom, dom)
        if dom in roofs:
            rf = roofs[dom]
            rec["dominant"] = {k: rf[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us") if k in rf}
            if "classes" in rf:             # (meta-training: every launch class with its share of the step)
                rec["dominant"]["classes"] = {k: {kk: v[kk] for kk in ("ms_per_step", "share", "tflops") if kk in v} for k, v in rf["classes"].items()}
                rec["dominant"]["whole_step"] = rf.get("whole_step")
        if share:
            rec["launch_time_share"] = share
        rec["other_rooflines"] = {k: {"frac": v.get("frac"), "achieved": v.get("achieved"), "unit": v.get("unit", "")[:8]} for k, v in roofs.items() if k != dom}
        if j.get("mean_acc") is not None:
            rec["mean_acc"] = j["mean_acc"]
        rec["profile"] = "profiles/r06_other_configs_%s_kernel_stats.txt (rocprofv3 --kernel-trace --stats of this command, tools/r06_other_configs_profiles.sh)" % key
        out[key] = rec
    return out


def default_line_with_other_configs(args):
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--no-other-configs"], stdout=subprocess.PIPE, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        sys.stdout.write(r.stdout)
        return r.returncode or 1
    out = json.loads(line[-1])
    out["other_configs"] = other_configs_children(args)
    print(json.dumps(out))
    return 0


def rank_census(dist, dev):
    """Who took part in this line: every rank reports (rank, device index, device name, PCI bus id, pid) through the SAME process
    group the timed collectives use (all_gather_object) -- a --gpus N line whose census lists N distinct devices proves N RCCL ranks
    (round-5 verdict: no N > 1 run has ever been recorded).  Single process: one entry, no collective."""
    import torch as _t
    idx = _t.cuda.current_device()
    props = _t.cuda.get_device_properties(idx)
    me = {"rank": 0 if dist is None else dist.get_rank(), "device": idx, "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
    if dist is None:
        return {"world_size": 1, "backend": None, "ranks": [me]}
    allr = [None] * dist.get_world_size()
    dist.all_gather_object(allr, me)
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
            "distinct_devices": len({(r["device"], r["pci_bus_id"]) for r in allr}), "ranks": allr}


def metatrain_roofline(model, opt, eps, step_s, loss_fn=None, k=1):
    """Roofline of the meta-training step (BASELINE configs[3]; round-4 verdict "missing 4"): three EAGER steps (the graphed step
    replays the same launches from one hipGraph, where nothing can be bracketed) with every C-ABI launcher call timed by a pair of
    HIP events on its own stream (_lib.LaunchTimer).  The step is a dense contraction (MFMA-bound): the
    dominant class of launches is reported against the fp32-MFMA peak with its algorithmic FLOPs, summed per launch from the C-ABI
    arguments (2 * output positions * Cout * KH * KW * Cin for every convolution / linear launch of the class: the backbone's 105
    images of 84x84 AND the head's linear layers over 7,440 pair rows -- SURVEY.md section 8(d) counts conv + linear); the other
    classes are listed by their share of the eager step's kernel time.  The four trunk.4 / trunk.5 forward launches and three
    stride-1 data-gradient launches that run on the bf16x3 kernels (round 5) are counted with their class, at algorithmic FLOPs."""
    from meta_fine_tuning_amd import _lib

    def conv_flops(name, a):
        """Algorithmic FLOPs of one convolution-family launch from its C-ABI arguments (None: not one)."""
        if name in ("mft_conv2d_nhwc", "mft_conv2d_nhwc_ksplit", "mft_conv2d_nhwc_x3"):
            return _conv_fl(a, 6, 7, 8, 9, 10, 11, 12, 13, 14)
        if name == "mft_conv2d_wgrad_oihw_multi":                   # (the job array itself is the first argument: ops.WgradBatch.flush)
            return sum(2.0 * j.n_img * _osz(j.H, j.KH, j.stride, j.pad) * _osz(j.W, j.KW, j.stride, j.pad) * j.Cout * j.KH * j.KW * j.Cin
                       for j in a[0])
        if name.startswith(("mft_conv2d_dgrad_nhwc", "mft_conv2d_wgrad_nhwc", "mft_conv2d_wgrad_oihw")):
            return _conv_fl(a, 5, 6, 7, 8, 9, 10, 11, 12, 13)       # (n, H, W = the forward input's; Cin, Cout the forward's)
        return None

    def klass(name, backward):
        if name.startswith("mft_conv2d_dgrad") or (name == "mft_conv2d_nhwc_x3" and backward):
            return "data gradients"                 # (stride-1 3x3 data gradients of the large layers run as bf16x3 convolutions)
        if name.startswith("mft_conv2d_wgrad"):
            return "weight gradients"
        if name.startswith("mft_conv2d_nhwc"):
            return "forward convolutions"
        if name.startswith("mft_pair_"):
            return "pair-MLP (GNN Wcompute) forward + backward"
        if name.startswith("mft_bn_") or "bn_backward" in name:
            return "BatchNorm statistics / apply / backward"
        if name.startswith(("mft_adam", "mft_pack_", "mft_split_")):
            return "outer Adam + weight repack / plane refresh"
        return "other (pooling, losses, copies, GNN glue)"

    reps = 3
    per = {}
    for i in range(reps):
        with _lib.LaunchTimer(keep_args=True) as lt:
            opt.zero_grad()
            loss = (loss_fn or model.set_forward_loss)(eps[i % len(eps)])
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            calls = lt.collect(calls=True)
            lt.close()
        backward = False
        for name, ms, a in calls:
            backward = backward or ("backward" in name) or name.startswith(("mft_conv2d_dgrad", "mft_conv2d_wgrad"))
            c = per.setdefault(klass(name, backward), {"ms_per_step": 0.0, "launches_per_step": 0, "gflop": 0.0})
            c["ms_per_step"] += ms / reps
            c["launches_per_step"] += 1
            fl = conv_flops(name, a)
            if fl is not None:
                c["gflop"] += fl / reps / 1e9
    total = sum(c["ms_per_step"] for c in per.values())
    for c in per.values():
        c["launches_per_step"] //= reps
        c["share"] = round(c["ms_per_step"] / total, 3)
        c["ms_per_step"] = round(c["ms_per_step"], 4)
    flops = {k: v["gflop"] * 1e9 for k, v in per.items() if v["gflop"] > 0}
    dom = max(flops, key=lambda k: per.get(k, {"ms_per_step": 0.0})["ms_per_step"])
    ach = flops[dom] / (per[dom]["ms_per_step"] * 1e-3) / 1e12
    whole = k * 112e9 / step_s / 1e12           # 112 GFLOP per 84x84 meta-train episode (k episodes per step)
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": (PEAK_F32_MFMA / 1e12), "unit": "TFLOP/s", "frac": round(ach / (PEAK_F32_MFMA / 1e12), 4),
            "traffic": None, "kernel": "%s (fp32 MFMA implicit GEMM; %.1f algorithmic GFLOP per step in %d launches, summed per launch from the C-ABI arguments)"
                                       % (dom, flops[dom] / 1e9, per[dom]["launches_per_step"]),
            "classes": {k: (dict(v, gflop=round(v["gflop"], 2), tflops=round(flops[k] / (v["ms_per_step"] * 1e-3) / 1e12, 2)) if k in flops
                            else {kk: vv for kk, vv in v.items() if kk != "gflop"})
                        for k, v in sorted(per.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "eager_kernel_ms_per_step": round(total, 3),
            "whole_step": {"algorithmic_gflop": 112.0 * k, "tflops": round(whole, 2), "frac_of_f32_mfma_peak": round(whole / (PEAK_F32_MFMA / 1e12), 4),
                           "ms_per_step_graphed": round(step_s * 1e3, 3)},
            "method": "HIP events around every launcher call of three eager steps, each on the launcher's own stream (_lib.LaunchTimer); "
                      "kernel-trace cross-check: profiles/r05_metatrain_kernel_trace.txt"}


def bench_metatrain(args, rank, world, dev, dist):
    """BASELINE configs[3]: episode-parallel meta-training, 5-way 5-shot, 16 queries (105 images of 84x84 per episode and
    rank); a step = loss + full backward on HIP, ONE flat fp32 all-reduce of all 5.3 M gradients over RCCL, fused outer Adam
    (train.py:28, meta_template.py:76-92; parallel.FlatGradBucket).  Extra measurement, not the headline metric."""
    from meta_fine_tuning_amd import optim, parallel, synthetic
    from meta_fine_tuning_amd.io_utils import model_dict
    from meta_fine_tuning_amd.methods.gnnnet import GnnNet
    model = GnnNet(model_dict["ResNet10"], n_way=5, n_support=5).cuda()
    model.load_state_dict(synthetic.gnnnet_state_dict(seed=0))
    model.train()
    model.n_query = 16
    opt = optim.Adam(model.parameters())
    bucket = parallel.FlatGradBucket(model.parameters())
    # Episode source (BASELINE configs[3]: "miniImageNet-shaped synthetic").  --train-source fixed (default; rounds 1-4): eight
    # pre-made fp32 episodes cycled.  --train-source pool: every step SAMPLES its episode inside the timed region from a
    # miniImageNet-shaped uint8 pool resident in HBM -- randperm(64)[:5] classes, 21 distinct images per class, the training-side
    # transform in one mft_augment_views launch (train.ResidentEpisodeLoader); the pool itself is generated before the clock starts.
    if args.train_source == "pool":
        from meta_fine_tuning_amd import train as _train
        pool = synthetic.class_pool_u8("miniImageNet", torch.device(dev), seed=0)
        loader = _train.ResidentEpisodeLoader(pool, 5, 5, 16, 84, n_episode=1 << 30, aug=args.train_aug, seed=100 + rank)

        class _Eps:              # the loader's episodes, indexed like the fixed list
            def __getitem__(self, i):
                return loader.episode(0, i)

            def __len__(self):
                return 1 << 30
        eps = _Eps()
    else:
        eps = [synthetic.train_episode(5000 + 100 * rank + i, 5, 5, 16, 84).to(dev) for i in range(8)]

    from meta_fine_tuning_amd import graph_step
    finetune = args.workload == "metafinetune"                        # train.py --fine_tune: set_forward_loss_finetune (gnnnet.py:106-231)
    kk = max(1, args.episodes_per_rank)
    if kk > 1:
        assert not finetune and args.train_source == "fixed"
        base = eps
        eps = [torch.stack([base[(i + j) % len(base)] for j in range(kk)]) for i in range(len(base))]      # k pre-made episodes per step
    loss_fn = model.set_forward_loss_finetune if finetune else (model.set_forward_loss_lockstep if kk > 1 else model.set_forward_loss)
    graphed = graph_step.for_loop(model, loss_fn)                      # the episode loop's own path (MetaTemplate._episode_loop)
    if finetune:
        np.random.seed(10 + rank)

    def step(i):
        if graphed is not None:
            loss = graphed(eps[i % len(eps)])                          # forward + backward: one hipGraph replay after 3 eager steps
        else:
            opt.zero_grad()
            loss = loss_fn(eps[i % len(eps)])
            loss.backward()
        opt.grad_scale = 1.0 / bucket.allreduce_sum()                  # SUM over the ranks; 1 / W applied inside the fused Adam launch
        opt.step()
        return loss

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 4 if graphed is not None else 0)):      # (the capture happens on the 4th step)
        step(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    census = rank_census(dist, dev)
    roof = cpu = None
    if rank == 0 and not finetune:
        roof = metatrain_roofline(model, opt, eps, dt / args.steps, loss_fn, kk)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_subprocess(args.gen_examples, workload="metatrain")
    if rank == 0:
        print(json.dumps({
            "metric": "episodes/sec at 5-way N-shot (ResNet10+GNN) per GPU and 1/2/4/8-GPU node", "value": round(world * kk * args.steps / dt, 3),
            "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("meta-fine-tuning training step (train.py --fine_tune, gnnnet.py:106-231: 105 inner Adam steps on trunk.7 per "
                                    "episode, then the outer step)" if finetune else "meta-training step (BASELINE configs[3])") +
                                   ": 5-way 5-shot, 16 queries, 84x84, %s per rank and step, "
                                   "flat 21.2 MB gradient all-reduce + fused outer Adam" % ("one episode" if kk == 1 else "%d episodes in lockstep (opt-in --episodes-per-rank: per-episode BatchNorm statistics, gradients averaged over them: the update of a %d-rank episode-parallel run)" % (kk, kk * world)),
                       "episodes_per_step": kk * world,
                       "episode_source": ("sampled per step inside the timed region from a miniImageNet-shaped resident uint8 pool "
                                          "(64 x 600 x 84x84; %s transform on the device)" % ("--train_aug" if args.train_aug else "Resize + CenterCrop")
                                          if args.train_source == "pool" else "eight pre-made fp32 episodes, cycled"),
                       "parallelism": "episode-parallel x%d" % world},
            "last_loss": round(float(loss.detach().cpu()), 4), "graphed": graphed is not None and graphed.graph is not None,
            "rccl_ranks": census, "roofline": roof, "cpu_baseline": cpu}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--episodes-per-batch", type=int, default=None,
                    help="lockstep batch E (default: 128; 96 at --n-shot 20 and 64 at --n-shot 50, where the stem cache fits: "
                         "profiles/r05_c_other_configs.txt; the MFT_EPB environment variable overrides the default)")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--gen-examples", type=int, default=17)
    ap.add_argument("--n-shot", type=int, default=5, help="5 = BASELINE configs[1] (the metric); 20 = configs[2]; 50 = configs[4] "
                    "(compressed-GNN fold, finetune_50.py).  Non-default shots are extra measurements, not the headline line.")
    ap.add_argument("--device-aug", action="store_true",
                    help="generate the 2+G views on the GPU from resident uint8 64x64 (EuroSAT-shaped) source images inside the timed "
                         "region (mft_augment_views) instead of ingesting pre-made fp32 views; extra measurement")
    ap.add_argument("--workload", default="finetune", choices=["finetune", "metatrain", "metafinetune"],
                    help="finetune = BASELINE configs[1] (the metric, default); metatrain = configs[3]: one meta-training episode per "
                         "rank per step (set_forward_loss -> full backward -> flat-bucket RCCL all-reduce -> fused outer Adam)")
    ap.add_argument("--image-size", type=int, default=84, help="84 = BASELINE configs (the metric); 224 = the reference's hard-coded "
                    "image_size (train.py:72, finetune.py:429) -- extra measurement, FLOP-derived fields then refer to 84")
    ap.add_argument("--train-source", default="fixed", choices=["pool", "fixed"],
                    help="--workload metatrain / metafinetune: fixed = cycle eight pre-made fp32 episodes resident in HBM (default: synthetic "
                         "data generation excluded, SURVEY section 8(d), as the headline workload does); pool = sample every episode on the "
                         "device from a miniImageNet-shaped resident uint8 pool INSIDE the timed region (4.00 vs 3.87 ms per step)")
    ap.add_argument("--episodes-per-rank", type=int, default=1,
                    help="--workload metatrain: k episodes per optimizer step in lockstep on each GPU (GnnNet.set_forward_loss_lockstep: "
                         "per-episode BatchNorm statistics, gradients averaged over the k episodes = the update of a k-rank "
                         "episode-parallel run); value counts k episodes per step.  Opt-in, its own line; 1 = the reference's loop")
    ap.add_argument("--train-aug", action="store_true", help="with --train-source pool: the --train_aug transform "
                    "(RandomResizedCrop + ImageJitter + flip) instead of Resize + CenterCrop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default line only: skip the short child runs of BASELINE configs[2] / [3] / [4] and of 224x224 (`other_configs`)")
    ap.add_argument("--no-standalone", action="store_true", help="skip the extra standalone launches of the dominant kernel "
                    "(roofline.standalone); used for the PMC passes so that per-launch counter means cover the step mix only")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="(internal) time the CPU oracle on a bounded sample and print its JSON object; never touches the GPU")
    ap.add_argument("--no-prefetch", action="store_true", help="do not enqueue the next batch's ingest + stem cache beside "
                    "the current inner loop (A/B)")
    ap.add_argument("--no-defer-final", action="store_true", help="run each batch's final pass synchronously (A/B)")
    ap.add_argument("--no-pipeline", action="store_true", help="single-stream inner loop (A/B against the 2-stream pipeline)")
    ap.add_argument("--strong-episodes", type=int, default=600,
                    help="after the timed (weak-scaling) region: the reference's fixed 600-episode evaluation split over the ranks, "
                         "wall time end to end -> `strong_scaling` in the JSON line (0 = off; default config only)")
    ap.add_argument("--strong-only", action="store_true", help="(internal) run only the fixed-job strong-scaling leg and print its record")
    ap.add_argument("--emulate-world", type=int, default=8,
                    help="with the strong-scaling leg at --gpus 1: also run rank 0's share of the fixed job at this world size alone on "
                         "the one GPU (fresh process) and report 600 / its wall as the projected W-GPU figure (0 = off)")
    ap.add_argument("--validate-episodes", type=int, default=128,
                    help="self-validation: the first V slots of the resident pool are the first V episodes of the accuracy golden "
                         "G9 (tests/golden/g9_accuracy.npz, the reference's own finetune() at this configuration); before the "
                         "warm-up one batch is run on the golden's numpy permutation stream and its per-episode accuracies are "
                         "compared with the reference's (0 = off)")
    args = ap.parse_args()
    if args.episodes_per_batch is None:
        args.episodes_per_batch = int(os.environ.get("MFT_EPB", {20: 96, 50: 64}.get(args.n_shot, 128)))

    if args.cpu_baseline_only:
        import meta_fine_tuning_amd  # noqa: F401
        from meta_fine_tuning_amd import synthetic
        state = synthetic.gnnnet_state_dict(seed=0)
        if args.workload == "metatrain":
            cores = host_threads()
            r = cpu_baseline_c1(state, cores, reps=8)
            print(json.dumps({"value": round(1.0 / r["seconds"], 4), "unit": "episodes/s", "cores": cores, "kind": "port",
                              "sample": "%d meta-training steps (one 5-way 5-shot 16-query 84x84 episode each: set_forward_loss + "
                                        "backward + Adam over 104 tensors; median %.3f s after one warm-up) on the torch-CPU oracle "
                                        "with %d threads (os.cpu_count()=%s)" % (r["steps_timed"], r["seconds"], cores, os.cpu_count())}))
            return
        ep = synthetic.test_episode(2000, 5, 5, 15, 84, gen_examples=args.gen_examples)
        print(json.dumps(cpu_baseline(state, ep)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process never touches a GPU; it starts the N ranks as fresh children
        # (python -m torch.distributed.run, one rank per GPU over RCCL), relays their output -- rank 0 prints the ONE JSON
        # line -- and exits with their status
        raise SystemExit(self_launch(args.gpus))
    if (args.gpus == 1 and "RANK" not in os.environ and not args.no_other_configs and not args.strong_only
            and args.workload == "finetune" and args.strong_episodes > 0 and args.n_shot == 5 and args.image_size == 84 and args.epochs == 5
            and args.gen_examples == 17 and not args.device_aug):
        # the default line: this process never touches the GPU.  It runs the headline measurement as a child (the whole path below,
        # with --no-other-configs), THEN the short children of the other BASELINE configurations -- each alone on the GPU, none
        # before the headline (run first they cost it 7 %: 82.7 instead of 89.7 episodes/s, measured, through the package's thermal
        # state; run beside a parent that still holds its engine the 50-shot child does not get its stem caches) -- and prints the
        # ONE line: the headline child's, with `other_configs` added.
        raise SystemExit(default_line_with_other_configs(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    strong = others = None
    if (not args.strong_only and args.workload == "finetune" and args.strong_episodes > 0 and args.n_shot == 5 and args.image_size == 84
            and args.epochs == 5 and args.gen_examples == 17 and not args.device_aug):
        strong = strong_scaling_child(args, rank, world)          # before this process initialises the GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no CPU fallback exists for the product path")
    if world > 1:
        torch.set_num_threads(max(1, host_threads() // world))     # W ranks share the host: no oversubscribed OpenMP pools
    else:
        torch.set_num_threads(min(torch.get_num_threads(), host_threads()))     # (torch sizes its pool by os.cpu_count(), not by the cgroup quota)
    if os.environ.get("MFT_BENCH_ONE_DEVICE"):          # test hook: run W ranks on one GPU (with MFT_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    # MFT_FORCE_COLLECTIVES=1 with RANK / MASTER_* set: ONE rank still forms an RCCL process group and runs every collective of the
    # N-rank path (timing all-reduce, accuracy all-gather, barriers) -- what a one-GPU box can execute of `--gpus N` (tests)
    forced = world == 1 and os.environ.get("MFT_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ
    if world > 1 or forced:
        import torch.distributed as dist
        backend = os.environ.get("MFT_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    import meta_fine_tuning_amd  # noqa: F401
    from meta_fine_tuning_amd import engine as eng
    from meta_fine_tuning_amd import ops, synthetic

    if os.environ.get("MFT_WGRAD_TILE"):
        from meta_fine_tuning_amd import _lib
        _lib.lib().mft_debug_set_conv_tile(1000 + int(os.environ["MFT_WGRAD_TILE"]))
    for knob in os.environ.get("MFT_CONV_KNOBS", "").split(","):          # A/B hook: mft_debug_set_conv_tile codes
        if knob.strip():
            from meta_fine_tuning_amd import _lib
            _lib.lib().mft_debug_set_conv_tile(int(knob))
    for knob in os.environ.get("MFT_X3_KNOBS", "").split(","):            # same for mft_debug_set_x3_tile
        if knob.strip():
            from meta_fine_tuning_amd import _lib
            _lib.lib().mft_debug_set_x3_tile(int(knob))
    if args.workload in ("metatrain", "metafinetune"):
        return bench_metatrain(args, rank, world, dev, dist)
    if args.strong_only:
        rec = strong_scaling_leg(args.strong_episodes, g9_state(), rank, world, dev, e_max=args.episodes_per_batch,
                                 emulate_world=args.emulate_world if (world == 1 and "--emulate-world" in sys.argv) else 0)
        if rank == 0:
            print(json.dumps(rec))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    E = args.episodes_per_batch
    n_way, n_shot, n_query, size = 5, args.n_shot, 15, args.image_size
    if n_shot != 5 or size != 84:
        args.no_cpu_baseline = True
    views = 2 + args.gen_examples
    # G9's weights (seeded backbone + head meta-trained with the reference): accuracies are then comparable with the golden
    validate = (args.validate_episodes if (n_shot == 5 and size == 84 and args.epochs == 5 and args.gen_examples == 17
                                            and not args.device_aug and rank == 0) else 0)
    validate = min(validate, E)
    state = g9_state() if (n_shot == 5) else synthetic.gnnnet_state_dict(seed=0)
    e = eng.FinetuneEngine(state, n_way, n_shot, n_query, size, n_views=views, fine_tune_epoch=args.epochs,
                           episodes_per_batch=E, device=dev, pipeline=not args.no_pipeline, fold50=(n_shot == 50))
    # resident synthetic episodes (class-structured so accuracy is meaningful); distinct per rank
    if args.device_aug:
        from meta_fine_tuning_amd import augment
        g = torch.Generator(device=dev)
        g.manual_seed(77 + rank)
        srcs = [torch.randint(0, 256, (n_way, n_shot + n_query, 64, 64, 3), generator=g, device=dev, dtype=torch.uint8)
                for _ in range(E)]
        aug_rs = np.random.RandomState(1234 + rank)
        pool = None
    else:
        head = g9_episodes(validate, dev, args.gen_examples) if validate else []
        pool = head + [synthetic.test_episode_device(1000 * 2 + rank * 100000 + i, dev, n_way, n_shot, n_query, size,
                                                     gen_examples=args.gen_examples, noise=G9_NOISE if n_shot == 5 else 1.0)
                       for i in range(len(head), E)]

    def one_batch():
        if args.device_aug:        # fresh random views every batch, parameters drawn on the host inside the timed region
            eps = [(srcs[i], augment.sample_view_params(aug_rs, n_way * (n_shot + n_query), 64, 64, size, args.gen_examples))
                   for i in range(E)]
            return e.run_batch(eps, sources=True, defer_final=not args.no_defer_final)
        # the final pass + GNN of a batch is enqueued on a third stream and overlaps the next batch's ingest / first steps;
        # everything is complete before the closing barrier + device synchronisation of the timed region
        return e.run_batch(pool, defer_final=not args.no_defer_final, prefetch=None if args.no_prefetch else pool)

    y_query = np.repeat(np.arange(n_way), n_query)
    validation = None
    if validate:
        # the golden's permutation stream: np.random.seed(10), five permutations of 500 per episode, episode by episode
        # (finetune.py:425,272) -- slot i of this batch runs exactly what the reference ran for episode i
        gz = np.load(os.path.join(ROOT, "tests", "golden", "g9_accuracy.npz"))
        assert list(gz["cfg_B"][:2]) == [5, 17] and int(gz["cfg_B"][2]) >= validate
        np.random.seed(10)
        sc = e.run_batch(pool)
        got = (sc[:validate].argmax(2).cpu().numpy() == y_query[None]).mean(1) * 100.0
        ref = gz["acc_B"][:validate]
        validation = {"episodes": int(validate), "mean_acc": round(float(got.mean()), 3), "golden_mean_acc": round(float(ref.mean()), 3),
                      "abs_diff": round(abs(float(got.mean()) - float(ref.mean())), 3),
                      "episodes_identical": int((np.abs(got - ref) < 1e-9).sum()),
                      "episodes_within_2_queries": int((np.abs(got - ref) <= 2 * 100.0 / 75 + 1e-9).sum()),
                      # north_star: +-0.2 % on the 600-episode mean; V episodes carry (600 / V)^0.5 times the sampling spread
                      "bar": round(0.2 * (600.0 / validate) ** 0.5, 3),
                      "ok": bool(abs(float(got.mean()) - float(ref.mean())) <= 0.2 * (600.0 / validate) ** 0.5),
                      "what": "first %d episodes of tests/golden/g9_accuracy.npz config B (the reference's finetune() on the same "
                              "episodes, weights and numpy stream): fp32 implementations agree per episode up to Adam sign flips "
                              "(DESIGN.md section 6)" % validate}
    np.random.seed(10 + rank)

    # ---- kernel timers: HIP events recorded on the stream each launcher enqueues on (the engine runs the frozen trunk and the
    # last-block half of a step on two different HIP streams), taken at the C-ABI by _lib.LaunchTimer -- no patched functions;
    # the algorithmic work of a launch is derived from the launcher's own arguments (include/mft_hip.h).  Dominant kernel of the
    # path = the fused weight-gradient + Adam kernel (HBM-bound: w, m, v of every episode are read and written once per step);
    # the implicit-GEMM convolutions (MFMA-bound) are reported next to it.
    from meta_fine_tuning_amd import _lib as _L

    def timed_launches(fn):
        """Run ``fn`` with an event pair around every launch of the families above -> {family: [(ms, work), ...]}."""
        with _L.LaunchTimer(only=LAUNCH_WORK.__contains__, keep_args=True) as lt:
            fn()
            torch.cuda.synchronize()
            calls = lt.collect(calls=True)
            lt.close()
        fam = {"adam": [], "f32": [], "x3": []}
        for name, ms, a in calls:
            kind, work = LAUNCH_WORK[name]
            fam[kind].append((ms, float(work(a))))
        return fam

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    accs = []
    for _ in range(args.warmup):
        one_batch()
    sync_all()
    power = PowerSampler() if rank == 0 else None
    if power is not None:
        power.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc = one_batch()
        accs.append(sc)
    sync_all()
    dt = time.perf_counter() - t0
    power = power.stop() if power is not None else None
    cdev = "cpu" if (dist is not None and dist.get_backend() == "gloo") else dev        # gloo (test hook) gathers host tensors
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    acc = torch.cat(accs).argmax(2).cpu().numpy() == y_query[None]
    acc_ep = acc.mean(1) * 100.0
    if dist is not None:
        gathered = [torch.zeros(len(acc_ep), device=cdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(acc_ep, device=cdev, dtype=torch.float64))
        acc_ep = torch.cat(gathered).cpu().numpy()

    # ---- rooflines: one more identical batch (same two-stream pipeline) with an event pair around every launch of the
    # two kernel families.  In-situ durations include the slowdown from the kernel co-running on the other stream; the
    # rocprofv3 kernel-trace of this same command (profiles/) must and does show the same averages.
    roof = roof_mfma = roof_x3 = None
    if rank == 0:
        fam = timed_launches(one_batch)
        adam_events, conv_events, x3_events = fam["adam"], fam["f32"], fam["x3"]
        a_ms = sum(t for t, _ in adam_events)
        a_by = sum(f for _, f in adam_events)
        n_a = len(adam_events)
        ach = a_by / (a_ms * 1e-3) / 1e9
        big = [(t, f) for t, f in adam_events if f == max(x[1] for x in adam_events)]
        roof = {"bound": "hbm", "kernel": "wgrad_adam_fwd_kernel (trunk.7 weight gradient with torch.optim.Adam fused in the epilogue AND "
                                          "the next inner step's convolution from the weight tiles just updated; per-episode w,m,v "
                                          "streamed once per inner step and not read again by a forward launch; MFT_FUSE_NEXT=0: "
                                          "wgrad_adam_rows_kernel + separate forward launches)" if e.fused_last_loop else
                                          "wgrad_adam_rows_kernel (trunk.7 weight gradient with torch.optim.Adam fused in the "
                                          "epilogue; per-episode w,m,v streamed once per inner step)",
                "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                "traffic": pmc_traffic(E, e.fused_last_loop), "launches": n_a, "avg_launch_us": round(a_ms * 1e3 / n_a, 2),
                "algorithmic_mb_per_launch": round(a_by / n_a / 1e6, 2),
                "largest_shape": {"what": "trunk.7.C2 (512x512x3x3) x %d episodes" % E,
                                  "avg_launch_us": round(sum(t for t, _ in big) * 1e3 / len(big), 2),
                                  "achieved": round(big[0][1] * len(big) / (sum(t for t, _ in big) * 1e-3) / 1e9, 1)},
                "practical_ceiling_note": "a pure 3-read/3-write Adam stream reaches 4.9-6.4 TB/s on this part depending on WHERE "
                                          "its three arrays live (tools/placement_scan.py); the engine picks the slabs' buffers by "
                                          "measured rate (slab_placement), stream_reference.on_engine_slabs is that stream over the "
                                          "engine's own slabs; in situ the step runs at the package power limit with the shader "
                                          "clock throttled (see power)"}
        # the same kernel with the GPU to itself (no trunk stream beside it): what the overlap costs the HBM-bound launch
        try:
            if args.no_standalone:
                raise RuntimeError("skipped (--no-standalone)")
            gen = torch.Generator(device=dev)
            gen.manual_seed(5)
            xs = torch.randn(E * 5, 3, 3, 512, device=dev, generator=gen)
            dys = torch.randn(E * 5, 3, 3, 512, device=dev, generator=gen) * 1e-3
            # on the engine's own trunk.7.C2 slabs (where w / m / v live decides the rate: engine.AdaptState._place); every
            # result of the run is already on the host, so they may be overwritten
            ws, ms_, vs_ = e.adapt.w.c2w, e.adapt.m.c2w, e.adapt.v.c2w
            ws.normal_(0.0, 0.02, generator=gen); ms_.zero_(); vs_.zero_()
            ops.conv2d_wgrad_adam(xs, dys, ws, ms_, vs_, 512, 3, 3, 1, 1, 1, 5)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(5):
                ops.conv2d_wgrad_adam(xs, dys, ws, ms_, vs_, 512, 3, 3, 1, 1, 2 + it, 5)
            e1.record()
            torch.cuda.synchronize()
            t_us = e0.elapsed_time(e1) * 1e3 / 5
            roof["standalone"] = {"what": "trunk.7.C2 x %d episodes on the engine's own slabs, no co-running stream, gradient + Adam only "
                                          "(wgrad_adam_rows_kernel)" % E, "avg_launch_us": round(t_us, 2),
                                  "achieved": round(24.0 * ws.numel() / (t_us * 1e-6) / 1e9, 1)}
            if e.fused_last_loop:
                # the fused form alone: same update + the next step's C2 forward, BatchNorms, add, ReLU, pool from the updated tiles
                tn = {k_: torch.empty((E * 5, 3, 3, 512), device=dev) for k_ in ("c2", "out", "sc")}
                st_ = {k_: torch.empty((E, 512), device=dev) for k_ in ("m2", "s2", "ms", "ss")}
                tn["sc"].normal_(generator=gen)
                ft_ = torch.empty((E * 5, 512), device=dev)
                ad_ = e.adapt.w

                def fused_once(step_):
                    ops.wgrad_adam_next_forward(xs, dys, ws, ms_, vs_, 3, 3, 1, 1, step_, 5, x_next=xs, mode=ops.WF_EXIT, raw=tn["c2"], act=tn["out"],
                                    gamma=ad_.bn2g, beta=ad_.bn2b, gbs=512, mean=st_["m2"], rstd=st_["s2"], sc_raw=tn["sc"],
                                    gamma_s=ad_.bnsg, beta_s=ad_.bnsb, mean_s=st_["ms"], rstd_s=st_["ss"], pooled=ft_)
                fused_once(1)
                torch.cuda.synchronize()
                e0.record()
                for it in range(5):
                    fused_once(2 + it)
                e1.record()
                torch.cuda.synchronize()
                t_us = e0.elapsed_time(e1) * 1e3 / 5
                roof["standalone_fused"] = {"what": "the same launch WITH the next step's C2 forward + block exit (wgrad_adam_fwd_kernel), alone",
                                            "avg_launch_us": round(t_us, 2), "achieved": round(24.0 * ws.numel() / (t_us * 1e-6) / 1e9, 1)}
            del xs, dys
        except RuntimeError as ex:          # e.g. not enough free memory next to a large engine
            roof["standalone"] = {"error": str(ex)[:120]}
        # what a pure w/m/v stream gets on THIS lease (the same binary measures 5.2-6.3 TB/s on different leases): an Adam-shaped
        # 3-read / 3-write pass over scratch arrays of the size of one parameter slab, no gradient operand, no matrix work
        try:
            n_el = (E * 3673088) // 1024 * 1024
            sw, sm, sv = (torch.zeros(n_el, device=dev) for _ in range(3))
            st_ = ops._stream
            _L.check(_L.lib().mft_stream_probe(ops._p(sw), ops._p(sm), ops._p(sv), n_el, st_()), "mft_stream_probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _L.lib().mft_stream_probe(ops._p(sw), ops._p(sm), ops._p(sv), n_el, st_())
            e1.record()
            torch.cuda.synchronize()
            ceil_gbs = 24.0 * n_el * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            roof["stream_reference"] = {"what": "pure 3-read/3-write Adam-shaped stream over %.1f GB of scratch on this lease, no co-running work, "
                                              "no gradient operand (mft_stream_probe; a reference point, not a bound: rates of this probe vary 5.1-6.3 TB/s from lease to lease)" % (12.0 * n_el / 1e9), "achieved": round(ceil_gbs, 1), "unit": "GB/s",
                                      "frac_of_peak": round(ceil_gbs / PEAK_HBM_GBS, 4),
                                      "dominant_kernel_in_situ_vs_this": round(ach / ceil_gbs, 4)}
            del sw, sm, sv
            # the same probe over the engine's OWN w / m / v slabs (their placement is chosen by measured stream rate,
            # engine.AdaptState._place; the batch results are already on the host, so the slabs may be overwritten now)
            ad = e.adapt
            n_sl = ad.w.flat.numel() // 1024 * 1024
            _L.lib().mft_stream_probe(ops._p(ad.w.flat), ops._p(ad.m.flat), ops._p(ad.v.flat), n_sl, st_())
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                _L.lib().mft_stream_probe(ops._p(ad.w.flat), ops._p(ad.m.flat), ops._p(ad.v.flat), n_sl, st_())
            e1.record()
            torch.cuda.synchronize()
            slab_gbs = 24.0 * n_sl * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            roof["stream_reference"]["on_engine_slabs"] = {"achieved": round(slab_gbs, 1), "dominant_kernel_in_situ_vs_this": round(ach / slab_gbs, 4)}
        except RuntimeError as ex:
            roof["stream_reference"] = {"error": str(ex)[:120]}
        share_ms = {k: sum(t for t, _ in v) for k, v in fam.items()}
        tot_ms = sum(t for t, _ in conv_events)
        tot_fl = sum(f for _, f in conv_events)
        n_launch = len(conv_events)
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12
        if x3_events:
            x_ms = sum(t for t, _ in x3_events)
            x_fl = sum(f for _, f in x3_events)
            ach3 = x_fl / (x_ms * 1e-3) / 1e12
            n_prod = 3 if e.W.f16x2 else 6            # matrix-core flops executed per algorithmic flop
            roof_x3 = {"bound": "mfma", "kernel": ("conv_x3_kernel / conv_x3_s1_kernel / conv_x3_s1_bnin_kernel, NP = 2 (frozen trunk.4-6: fp32-accurate 3-term "
                                                   "f16x2 products on fp16 MFMA, two accumulators; bf16x3 / six products when MFT_TRUNK_F16X2=0 or the range guard "
                                                   "refuses the weights;") if e.W.f16x2 else "conv_x3_kernel / conv_x3_s1_kernel / conv_x3_s1_bnin_kernel (frozen trunk.4-6: fp32-accurate 6-term bf16x3 products on bf16 MFMA; BatchNorm statistics partials in the epilogue, BN1 + ReLU in C2's loader on trunk.4 / trunk.5)",
                       "achieved": round(ach3, 2), "peak": round(PEAK_BF16_MFMA / n_prod / 1e12, 1), "unit": "TFLOP/s (fp32-equivalent: "
                       "algorithmic 2*M*N*K; the kernel executes %d 16-bit MFMA flops per algorithmic flop, peak = 2500/%d; in situ these "
                       "launches share the GPU with the HBM-bound last-block stream and stretch to what it leaves -- see standalone)" % (n_prod, n_prod),
                       "products_per_flop": n_prod,
                       "frac": round(ach3 / (PEAK_BF16_MFMA / n_prod / 1e12), 4), "launches": len(x3_events),
                       "avg_launch_us": round(x_ms * 1e3 / len(x3_events), 2),
                       "vs_fp32_mfma_peak": round(ach3 / (PEAK_F32_MFMA / 1e12), 3)}
            if not args.no_standalone:
                # the same convolutions with the GPU to themselves: six trunk steps on one stream, no last-block stream beside them
                try:
                    prs = [eng.draw_perms(e.n_total, e.epochs, np.random.RandomState(900 + i)) for i in range(E)]
                    tabs = [t_ for t_ in e.step_tables(prs, E) if t_[0] == e.bs][:7]
                    e.trunk_step(torch.from_numpy(tabs[0][1]).to(dev), e.bs, 0)
                    torch.cuda.synchronize()
                    alone = timed_launches(lambda: [e.trunk_step(torch.from_numpy(t_[1]).to(dev), e.bs, 0) for t_ in tabs[1:]])["x3"]
                    s_ms = sum(t for t, _ in alone)
                    s_fl = sum(f for _, f in alone)
                    if alone and s_ms > 0:
                        sa = s_fl / (s_ms * 1e-3) / 1e12
                        roof_x3["standalone"] = {"what": "the same eight convolutions per step with no co-running last-block stream (6 trunk steps)",
                                                 "achieved": round(sa, 2), "frac": round(sa / (PEAK_BF16_MFMA / n_prod / 1e12), 4),
                                                 "vs_fp32_mfma_peak": round(sa / (PEAK_F32_MFMA / 1e12), 3),
                                                 "avg_launch_us": round(s_ms * 1e3 / len(alone), 2)}
                except Exception as ex:   # noqa: BLE001 -- an extra measurement must not cost the line
                    roof_x3["standalone"] = {"error": str(ex)[:120]}
        roof_mfma = {"bound": "mfma", "kernel": "conv_igemm_kernel + stem_conv_kernel (fp32 MFMA implicit GEMM: stem, "
                                                "weight-streaming per-episode trunk.7 launches, GNN GEMMs)",
                     "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                     "frac": round(achieved / (PEAK_F32_MFMA / 1e12), 4), "launches": n_launch,
                     "avg_launch_us": round(tot_ms * 1e3 / n_launch, 2),
                     "algorithmic_gflop_per_launch": round(tot_fl / n_launch / 1e9, 3)}

    placement = e.adapt.placement
    census = rank_census(dist, dev)

    if rank == 0:
        total_eps = E * args.steps * world
        value = total_eps / dt
        fl = episode_flops(n_way, n_shot, n_query, views, args.epochs)
        n_steps_ep = args.epochs * n_way * n_shot * (views + 1) // 5
        gb_step = 0.1122 - (0.0147 if e.fused_last_loop else 0.0)          # adaptable-state GB per episode and inner step
        out = {
            "metric": "episodes/sec at 5-way N-shot (ResNet10+GNN) per GPU and 1/2/4/8-GPU node",
            "value": round(value, 3), "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": ("all arithmetic is fp32-accurate: fp32 MFMA everywhere except the frozen trunk.4-6 convolutions, which split every "
                           "fp32 operand into two fp16 pieces (hi + 2^-11 lo, |residual| <= 2^-22) and accumulate the three leading piece "
                           "products in two fp32 accumulators (measured error vs float64: 0.45x the fp32-MFMA kernel's, "
                           "profiles/r04_a_f16x2_accuracy_and_time.txt; tests/test_kernels_gpu.py::test_conv2d_f16x2_is_fp32_accurate), and "
                           "the per-episode last-block forward / data-gradient launches, which use exact 3-way bf16 splits with six products"
                           if e.W.f16x2 else
                           "all arithmetic is fp32-accurate: fp32 MFMA everywhere except the frozen trunk.4-6 convolutions, which run as "
                           "exact 3-way bf16 splits with the six leading products accumulated in fp32 (error <= fp32 GEMM rounding; "
                           "tests/test_kernels_gpu.py::test_conv2d_bf16x3_is_fp32_accurate)"), "data": "synthetic",
            "config": {"workload": "5-way %d-shot ResNet10+GNN test-time finetune, 84x84, fine_tune_epoch=%d, "
                                   "gen_examples=%d (%d inner Adam steps/episode), 15 queries" %
                                   (n_shot, args.epochs, args.gen_examples, args.epochs * n_way * n_shot * (views + 1) // 5),
                       "episodes_per_step": E, "episodes_total": total_eps, "image_size": size,
                       "views": "generated on the GPU from uint8 64x64 sources inside the timed region" if args.device_aug
                       else "pre-made fp32 NCHW views resident in HBM",
                       "parallelism": "episode-parallel x%d" % world},
            "episode_tflop": round(fl / 1e12, 4),
            "whole_path_tflops": round(value * fl / 1e12, 2),
            "whole_path_frac_of_f32_mfma_peak": round(value * fl / world / PEAK_F32_MFMA, 4),
            "mean_acc": round(float(acc_ep.mean()), 2),
            "mean_acc_note": "all timed batches, resident pool re-run with fresh permutations each batch; weights = golden G9's "
                             "(meta-trained head), so this is a real accuracy, not chance" if n_shot == 5 else "seed-0 random head",
            "validation": validation,
            # whole path against the same HBM roof: the adaptable-state bytes an episode moves (per inner step: forward weights
            # 14.7 MB + data-gradient re-read 9.4 MB + Adam read/write of w, m, v 88.2 MB) over the wall time of
            # the timed region, everything else (trunk, final pass, ingest, host) counted as zero bytes
            # (with the fused next-step forward the forward's 14.7 MB weight read does not exist: 97.5 MB per step)
            "whole_path_hbm": {"algorithmic_gb_per_episode": round(gb_step * n_steps_ep, 2),
                               "per_inner_step_mb": round(gb_step * 1e3, 1),
                               "achieved": round(value / world * gb_step * n_steps_ep, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(value / world * gb_step * n_steps_ep / PEAK_HBM_GBS, 4),
                               "fused_next_forward": bool(e.fused_last_loop), "fused_next_forward_policy": bool(e.fuse_next)},
            "power": power if power is None else dict(power, joules_per_episode=round(power["socket_w_median"] * dt / (E * args.steps), 2)),
            "slab_placement": placement,
            "strong_scaling": strong,
            "rccl_ranks": census,
            "roofline": roof,
            "roofline_mfma": roof_mfma,
            "roofline_mfma_x3": roof_x3,
            # in-situ launch time of one batch by kernel family (ms, HIP events on each launcher's own stream; the two queues overlap,
            # so the sum exceeds the wall): which family a configuration is bound by
            "launch_time_share": {k: round(v, 2) for k, v in share_ms.items()},
            "other_configs": others,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_subprocess(args.gen_examples)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
