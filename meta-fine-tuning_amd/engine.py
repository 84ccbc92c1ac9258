"""Episode-batched test-time fine-tuning engine (the BASELINE.json hot path).

``finetune.finetune`` (finetune.py:182-328) fine-tunes ONE episode at a time: 500-5000 sequential
Adam steps on a 5-image mini-batch, i.e. ~2 GFLOP per step -- far too little to fill 256 CUs.
Episodes are independent (each reloads the checkpoint, finetune.py:185-198), so this engine runs
``E`` episodes in lockstep: step ``t`` of all episodes is one set of grouped launches over E*5
images with per-episode BatchNorm statistics (the frozen trunk.0-6 weights are shared) and
per-episode last-block weights / Adam state held in HBM (44 MB per episode).  Host work per batch
is index tables only; the numpy permutations are drawn in the reference's order.
"""
import os

import ctypes

import numpy as np
import torch

from . import functional as Fn
from . import ops
from . import settings


_DEBUG_SKIP_TRUNK = settings.current().debug_skip_trunk     # results are then WRONG; timing experiments only


def pick_slab_buffers(rates, K):
    """``rates``: {sorted triple of candidate indices: measured rate of the 3-read / 3-write stream over those three buffers}.
    Returns (w, m, v, w2): the pair (m, v) and the two weight buffers w, w2 that maximise min(rate(w, m, v), rate(w2, m, v)) --
    the deferred final pass alternates between two weight slabs that share the moment slabs."""
    import itertools
    best, best_score = None, -1.0
    for m_, v_ in itertools.combinations(range(K), 2):
        ws = sorted((rates[tuple(sorted((w_, m_, v_)))], w_) for w_ in range(K) if w_ not in (m_, v_))
        score = min(ws[-1][0], ws[-2][0])
        if score > best_score:
            best, best_score = (ws[-1][1], m_, v_, ws[-2][1]), score
    return best


_PLACEMENT_HINTS = {}
_SLAB_CANDIDATES = 12


class slab_candidates:
    """``with slab_candidates(8): FinetuneEngine(...)``: fewer candidate buffers for the slab-placement scan of engines built
    inside the block (C(8,3) = 56 triples instead of 220: 0.2 s instead of 0.8 s and 30 GB less transient memory) -- for short
    jobs such as one rank's share of a 600-episode evaluation, where the full scan costs what the better placement returns.
    ``slab_candidates(0)``: no scan at all (plain allocations) -- a job of ONE lockstep batch (one rank's 75 episodes of a
    600-episode evaluation on 8 GPUs runs for about a second: the scan costs more than any placement returns);
    ``slab_candidates(None)``: the default.  The MFT_SLAB_CANDIDATES environment variable still wins."""

    def __init__(self, k):
        self.k = k

    def __enter__(self):
        global _SLAB_CANDIDATES
        self.old = _SLAB_CANDIDATES
        if self.k is not None:
            _SLAB_CANDIDATES = int(self.k)

    def __exit__(self, *a):
        global _SLAB_CANDIDATES
        _SLAB_CANDIDATES = self.old


def _valid_hint(h, K=None):
    """A usable hint is a dict with four DISTINCT candidate indices (w, m, v, alternate w) and the two stream rates the scan
    measured for them.  Anything else -- a stale schema, a foreign or truncated file -- is "no hint"."""
    try:
        c = h["chosen"]
        if not (isinstance(c, (list, tuple)) and len(c) == 4 and all(type(i) is int and i >= 0 for i in c) and len(set(c)) == 4):
            return False
        if K is not None and max(c) >= K:
            return False
        return all(isinstance(h[k], (int, float)) and not isinstance(h[k], bool) and 0.0 < float(h[k]) < 1e6
                   for k in ("chosen_gbs", "chosen_alt_gbs"))
    except (KeyError, TypeError, ValueError):
        return False


def _hint_dir():
    """A directory only this user can write (0700): ~/.cache/mft, else a per-user directory in the temp directory that is
    verified to be OURS and not a symlink (a sticky shared /tmp is not trusted with a predictable file name)."""
    import stat
    import tempfile
    for d in (os.path.join(os.path.expanduser("~"), ".cache", "mft"), os.path.join(tempfile.gettempdir(), "mft-%d" % os.getuid())):
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            st = os.lstat(d)
            if stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o022):
                return d
        except OSError:
            continue
    return None


def _placement_hint(key, store=None, K=None):
    """Read (or, with ``store``, record) the slab-placement choice of an earlier scan: in-process dict first, then a JSON file
    in a per-user 0700 directory shared by the processes of one box.  Best effort -- any I/O problem, and any entry that does not
    pass ``_valid_hint``, just means "no hint" (the full scan runs)."""
    import json
    import tempfile
    d = _hint_dir()
    path = None if d is None else os.path.join(d, "slab_placement.json")
    if store is not None:
        _PLACEMENT_HINTS[key] = store
        if path is None:
            return store
        try:
            disk = {}
            try:
                with open(path) as f:
                    disk = json.load(f)
            except (OSError, ValueError):
                pass                                                               # (absent or broken: start over)
            if not isinstance(disk, dict):
                disk = {}
            disk[key] = store
            fd, tmp = tempfile.mkstemp(dir=d, prefix=".slab_placement.")          # O_EXCL, 0600, unpredictable name
            try:
                with os.fdopen(fd, "w") as f:
                    json.dump(disk, f)
                os.replace(tmp, path)
            except BaseException:
                try:
                    os.unlink(tmp)
                except OSError:
                    pass
                raise
        except (OSError, ValueError, TypeError):
            pass
        return store
    h = _PLACEMENT_HINTS.get(key)
    if h is None and path is not None:
        try:
            with open(path) as f:
                disk = json.load(f)
            h = disk.get(key) if isinstance(disk, dict) else None
        except (OSError, ValueError):
            h = None
    return h if _valid_hint(h, K) else None


def fuse_next_policy(setting, E, stem_cached, image_size, f16x2):
    """``setting``: "1" / "0" / "auto" (MFT_FUSE_NEXT) -> bool.  auto = where the fused next-step forward measured faster (round 4,
    FinetuneEngine._build): whole waves of the walking kernel (E % 32 == 0), a light trunk stream beside it (stem cache, f16x2
    convolutions), the 84 x 84 maps it was tuned on."""
    if setting in ("0", "1"):
        return setting == "1"
    return bool(E >= 32 and E % 32 == 0 and stem_cached and f16x2 and image_size <= 84)


_WF_XCD_SET = False


def _set_wgrad_fwd_xcd_once():
    """MFT_WF_XCD (A/B hook: workgroup -> XCD order of the fused launch) is process-global kernel state: read ONCE per process,
    not per engine -- an engine built later must not change the launch order of engines that already exist."""
    global _WF_XCD_SET
    if not _WF_XCD_SET:
        if not settings.current().wf_xcd:       # the default (one XCD per episode) is the library's own: the product path never touches
            ops._lib.lib().mft_wgrad_fwd_set_xcd(0)      # the hook (include/mft_hip_testing.h); only the A/B value does
        _WF_XCD_SET = True


class AdaptState:
    """Per-episode adaptable parameters + gradient + Adam moments (four tensor-major slabs)."""

    def __init__(self, E, device):
        self.E = E
        self.placement = None
        flats = self._place(E, device)
        self.w = Fn.LastBlockSlab(E, device, flat=flats[0])
        self.g = Fn.LastBlockSlab(E, device, flat=flats[3])
        self.m = Fn.LastBlockSlab(E, device, flat=flats[1])
        self.v = Fn.LastBlockSlab(E, device, flat=flats[2])
        self.w_alt_flat = flats[4]          # second weight slab of the deferred final pass (FinetuneEngine._flip_buffers), placed as well
        self.step = 0

    def _place(self, E, device):
        """WHERE the w / m / v slabs live decides how fast they stream: the same 3-read / 3-write pass measures 4.9-6.3 TB/s over
        different triples of separately allocated 1.9 GB buffers of ONE process, reproducibly per triple (tools/placement_scan.py:
        typically the first ~10 GB a process allocates are the slow ones among themselves) -- and these slabs are 3/4 of the bytes
        the inner loop moves.  So K candidate buffers are allocated (groups of four, ballast between the groups), every triple is
        timed with the Adam-shaped probe (mft_stream_probe; ~1.5 s in all at E = 128) and pick_slab_buffers chooses (w, m, v) and the
        second weight slab of the deferred final pass; the gradient slab takes any other candidate, the rest goes back to the
        driver.  Placement does not touch any result.  MFT_SLAB_CANDIDATES=0 turns it off."""
        total = E * Fn.ADAPT_NUMEL
        cfg = settings.current()
        K = _SLAB_CANDIDATES if cfg.slab_candidates is None else cfg.slab_candidates
        ballast_gb = min(cfg.slab_ballast_gb, 6.4 * total * 4 / (1 << 30))
        dev = torch.device(device)
        if K >= 5 and dev.type == "cuda" and total * 4 >= (64 << 20):
            free_b, _ = torch.cuda.mem_get_info(dev)
            budget = int(free_b * 0.4)
            K = min(K, budget // (total * 4))
            ballast_gb = min(ballast_gb, max(0.0, (budget - K * total * 4) / 2.0 / (1 << 30)))
        if K < 5 or dev.type != "cuda" or total * 4 < (64 << 20):
            return [None, None, None, None, None]
        import itertools
        with torch.cuda.device(dev):
            # candidates in groups of four with throw-away ballast between the groups: consecutive allocations share a region of
            # the address space, and regions of tens of GB differ (36 consecutive 1.9 GB buffers: triples of neighbours stream at
            # 4.9-5.4 TB/s over the first ~32 GB, 5.7-6.4 over the next ~20 GB, 5.3 after that)
            cands, ballast = [], []
            for i in range(K):
                if i and i % 4 == 0 and ballast_gb > 0:
                    ballast.append(torch.empty(int(ballast_gb * (1 << 28)), device=dev))
                cands.append(torch.empty(total, device=dev))
            for c in cands:
                c.zero_()
            del ballast
            n = total // 1024 * 1024
            lib = ops._lib.lib()

            def rate(t, reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    ops._lib.check(lib.mft_stream_probe(ops._p(cands[t[0]]), ops._p(cands[t[1]]), ops._p(cands[t[2]]), n,
                                                        ops._stream()), "mft_stream_probe")
                b.record()
                b.synchronize()
                return 24.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e9

            triples = list(itertools.combinations(range(K), 3))
            rate(triples[0], 2)                                        # clocks up
            # A process that allocates the same candidates in the same order tends to get the same physical placement, so the
            # choice of an earlier scan (this process, or another one on this box: a small JSON in the temp directory) is
            # re-used when ONE probe of its two triples confirms the rates it promised (within 3 %); otherwise the full scan
            # runs (1.5 s at E = 128 -- a quarter of a 600-episode evaluation's fixed cost).  MFT_SLAB_HINTS=0 turns it off.
            hint_key = "%s|%d|%d|%.2f" % (torch.cuda.get_device_name(dev), total, K, ballast_gb)
            hint = _placement_hint(hint_key, K=K) if cfg.slab_hints else None
            best = rates = None
            if hint is not None:
                w_, m_, v_, w2_ = hint["chosen"]
                r1, r2 = rate(tuple(sorted((w_, m_, v_))), 2), rate(tuple(sorted((w2_, m_, v_))), 2)
                if r1 >= 0.97 * hint["chosen_gbs"] and r2 >= 0.97 * hint["chosen_alt_gbs"]:
                    best = (w_, m_, v_, w2_)
                    self.placement = dict(hint, chosen_gbs=round(r1, 1), chosen_alt_gbs=round(r2, 1),
                                          source="hint of an earlier scan, confirmed by a probe of its two triples")
            if best is None:
                rates = {t: rate(t, 2) for t in triples}
                w_, m_, v_, w2_ = best = pick_slab_buffers(rates, K)
                self.placement = {"candidates": K, "chosen_gbs": round(rates[tuple(sorted((w_, m_, v_)))], 1),
                                  "chosen_alt_gbs": round(rates[tuple(sorted((w2_, m_, v_)))], 1), "best_gbs": round(max(rates.values()), 1),
                                  "worst_gbs": round(min(rates.values()), 1), "median_gbs": round(float(np.median(list(rates.values()))), 1),
                                  "first_three_allocations_gbs": round(rates[(0, 1, 2)], 1), "chosen": [w_, m_, v_, w2_],
                                  "source": "full scan of %d triples" % len(triples)}
                _placement_hint(hint_key, self.placement)
            rest = [i for i in range(K) if i not in best]
            keep = [cands[w_], cands[m_], cands[v_], cands[rest[0]], cands[w2_]]
            del cands
            torch.cuda.empty_cache()          # the unused candidates and the ballast go back to the driver, not into torch's cache
            return keep

    def reset(self, W):
        self.w.load_shared(W)
        self.m.flat.zero_()
        self.v.flat.zero_()
        self.step = 0


def draw_perms(n_total, total_epoch, rng=np.random):
    """finetune.py:270-272: one np.random.permutation(n_total) per epoch, from the global numpy RNG."""
    return [rng.permutation(n_total) for _ in range(total_epoch)]


def _on_device(fn):
    """Run an engine entry point with the engine's device current (launches, streams and events all belong to it)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with torch.cuda.device(self.dev):
            return fn(self, *a, **k)
    return wrapped


class FinetuneEngine:
    def __init__(self, state, n_way=5, n_support=5, n_query=15, image_size=84, n_views=19, fine_tune_epoch=5,
                 episodes_per_batch=16, batch_size=5, lr=0.01, device="cuda:0", head_state=None, fold50=False,
                 fused_adam=True, pipeline=True, stem_cache=True, mode="gnn", x3=True, graph=False, trunk_chunk=None,
                 fuse_next=None):
        """state: GnnNet state dict ('feature.*', 'fc.*', 'gnn.*'); n_views = 2 + gen_examples.
        ``head_state`` overrides the fc/gnn weights (the reference scores with the *loaded model*, finetune.py:316).
        ``mode`` "gnn": finetune.finetune (inner loss on the raw feature, GNN scoring);
        "linear": finetune.finetune_linear (finetune.py:45-174: per-episode Linear(512, n_way) classifier trained together
        with the last block over the ORIGINAL support images only; scores = softmax(classifier(features)))."""
        assert mode in ("gnn", "linear")
        self.mode = mode
        # ``graph``: capture one inner step (single stream) as a hipGraph and replay it for every step -- ~40 launches
        # become 3 (index copy, label copy, replay).  Measured: no gain (a step is a chain of dependent kernels bound by
        # per-kernel latency on the GPU, not by the host launch rate); off by default.
        self.use_graph = bool(graph) and mode == "gnn" and fused_adam
        if self.use_graph:
            pipeline = False
        self._graphs = {}
        self.fused_last_loop = False       # did the last inner_loop() run the fused weight-gradient + Adam + next-step-forward launches?
        self._dbg_x6 = {}
        self._alt = None
        self._pre = None            # ingest + stem cache of the NEXT batch, enqueued on their own stream (run_batch(prefetch=))
        self._pre_bufs = None
        if not torch.cuda.is_available():
            raise RuntimeError("FinetuneEngine needs an MI355X (HIP) device; there is no CPU fallback")
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self._raw_stream = None
        with torch.cuda.device(self.dev):
            self._build(state, n_way, n_support, n_query, image_size, n_views, fine_tune_epoch, episodes_per_batch, batch_size, lr,
                        head_state, fold50, fused_adam, pipeline, stem_cache, mode, x3, trunk_chunk)
            if fuse_next is not None:
                self.fuse_next = bool(fuse_next)

    def close(self):
        """Release what the caching allocator does not own: the raw HIP priority stream (mft_stream_create_priority)."""
        if self._raw_stream is not None:
            torch.cuda.synchronize(self.dev)
            ops._lib.lib().mft_stream_destroy(ctypes.c_void_p(self._raw_stream))
            self._raw_stream = None
            self.s_trunk = None
            if getattr(self, "_raw_last", None) is not None:
                ops._lib.lib().mft_stream_destroy(ctypes.c_void_p(self._raw_last))
                self._raw_last = None
                self.s_last = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _build(self, state, n_way, n_support, n_query, image_size, n_views, fine_tune_epoch, episodes_per_batch, batch_size, lr,
               head_state, fold50, fused_adam, pipeline, stem_cache, mode, x3, trunk_chunk):
        self.n_way, self.n_support, self.n_query = n_way, n_support, n_query
        self.size, self.n_views, self.epochs = image_size, n_views, fine_tune_epoch
        self.E, self.bs, self.lr = episodes_per_batch, batch_size, lr
        self.fold50 = fold50
        self.fused_adam = fused_adam
        self.n_per_view = n_way * n_support
        # finetune.py:214-233,269: view 0 twice + every other view; finetune_linear permutes support_size only (:139-141)
        self.n_total = self.n_per_view * (n_views + 1) if mode == "gnn" else self.n_per_view
        self.n_all = n_way * (n_support + n_query)
        fsd = {k[len("feature."):]: v for k, v in state.items()
               if k.startswith("feature.") and not k.startswith(("feature2.", "feature3."))}
        # the largest train-mode BatchNorm group this engine runs: the stem's, over the n_all images of the final pass (the inner
        # loop's groups are `batch_size` images) -- the fp16 range proof of the f16x2 trunk is made for that size
        self.W = Fn.ResNet10Weights(fsd, self.dev, x3=x3, max_rows=Fn.stem_rows_bound(max(self.n_all, batch_size), image_size))
        self.G = Fn.GnnHeadWeights(head_state if head_state is not None else state, self.dev, n_way) if mode == "gnn" else None
        if mode == "linear":
            self.cls = {k: torch.zeros((self.E, n_way, 512) if k.endswith("W") else (self.E, n_way), device=self.dev)
                        for k in ("W", "b", "mW", "vW", "mb", "vb")}
        self.arena = Fn.Arena(self.dev)
        self.arena_trunk = Fn.Arena(self.dev)      # the frozen-trunk stream owns its own buffers / BN workspace
        self.adapt = AdaptState(self.E, self.dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)     # device-side Adam step counter / bias
        self.hyper = torch.zeros(2, device=self.dev)                           # corrections (graph replay)
        self.pipeline = pipeline
        # The weight-gradient + Adam launches of step t can also run step t+1's last-block forward from the weight tiles they have
        # just updated (csrc/wgrad_fwd.hip, fuse_next): the updated weights are then not read back by a forward launch (7.64 ->
        # 6.64 parameter units of HBM traffic per step).  The launch is a WALKING kernel (one long-lived workgroup per (episode, 32
        # output channels), two per CU), bound by its per-tile chain rather than by HBM, so it pays only where (a) its E * 16
        # workgroups fill whole waves of the 512 resident slots (E a multiple of 32) and (b) the trunk stream beside it is light
        # (stem cache present; since round 4 the f16x2 trunk convolutions).  Same-lease A/B, round 4
        # (profiles/r04_b_fuse_next_*.txt): E = 128: 85.4 -> 87.1 episodes/s, E = 32: 59.4 -> 68.7, 20-shot E = 96: 20.3 -> 21.3;
        # but E = 120: 84.0 -> 80.3 (3.75 waves), 50-shot E = 128 (no stem cache): 7.93 -> 7.48, 224x224: 19.8 -> 19.6.
        # MFT_FUSE_NEXT = auto (default: that rule, fuse_next_policy, applied once the stem cache is decided) | 1 | 0.
        cfg = settings.current()
        self._fuse_next_setting = cfg.fuse_next      # "auto" | "1" | "0"; resolved to the boolean self.fuse_next below
        self.fuse_next = False
        _set_wgrad_fwd_xcd_once()
        # measured at E=128 (A/B in one session): steps per trunk launch set 1 / 2 / 4 / 8 -> 3.75 / 3.77 / 3.89 / 3.96 ms per
        # step: longer trunk launches disturb the HBM-bound stream more than they gain in efficiency; splitting one step's
        # trunk into 2 / 4 episode sub-batches gives 3.78 / 4.12: one step per launch set is the optimum
        # a single episode (the reference's per-episode finetune() call without the LookaheadLoader): the frozen trunk of 32 steps per
        # set of launches -- at E = 1 a trunk pass is 5 images and pure launch latency (5.4 -> 12.3 episodes/s, tools/small_e.py)
        self.trunk_chunk = (int(trunk_chunk) if trunk_chunk is not None else
                            cfg.trunk_chunk if cfg.trunk_chunk is not None else (32 if episodes_per_batch == 1 else 1))
        # Queue priorities: the last-block stream is the critical path (its 8 launches per step are serial and HBM-bound), the
        # trunk stream only has to stay one step ahead.  A/B (one session, two runs each): last high / trunk normal 68.1, 68.1;
        # trunk high / last normal 67.4; both normal 67.2, 67.2 episodes/s.
        # (trunk at the device's least priority, below torch's range: 68.2 / 68.1 / 67.7 vs 67.8 / 68.0 / 67.5 at normal)
        prio = cfg.trunk_priority
        self.s_trunk = None
        self._raw_last = None
        # CU partition (opt-in, MFT_TRUNK_CUS_PER_XCD = n): the frozen-trunk stream may use n of the 32 CUs of every XCD, the
        # last-block stream the other 32 - n (hipExtStreamCreateWithCUMask; mask bit i -> XCD i % 8, CU i // 8).  A 3-read /
        # 3-write stream keeps 93 % of its rate on 160 of the 256 CUs (tools/microbench/adam_cus.hip) while the matrix-bound
        # trunk scales with its CU count.
        n_tr = cfg.trunk_cus_per_xcd
        if pipeline and 0 < n_tr < 32:
            def masked(bits):
                words = (ctypes.c_uint * 8)()
                for b in bits:
                    words[b // 32] |= (1 << (b % 32))
                out = ctypes.c_void_p()
                ops._lib.check(ops._lib.lib().mft_stream_create_cumask(ctypes.cast(words, ctypes.c_void_p), 8, ctypes.byref(out)),
                               "mft_stream_create_cumask")
                return out.value
            self._raw_stream = masked([i for i in range(256) if (i // 8) >= 32 - n_tr])
            self.s_trunk = torch.cuda.ExternalStream(self._raw_stream, device=self.dev)
            if not cfg.trunk_cus_only:      # (=1: only the trunk is confined; the last block may run anywhere)
                self._raw_last = masked([i for i in range(256) if (i // 8) < 32 - n_tr])
                self.s_last = torch.cuda.ExternalStream(self._raw_last, device=self.dev)
        elif pipeline and prio > 0:              # below torch's range: a HIP stream at the device's least priority
            out = ctypes.c_void_p()
            with torch.cuda.device(self.dev):
                ops._lib.check(ops._lib.lib().mft_stream_create_priority(prio, ctypes.byref(out), None), "mft_stream_create_priority")
            self._raw_stream = out.value
            self.s_trunk = torch.cuda.ExternalStream(out.value, device=self.dev)
        elif pipeline:
            self.s_trunk = torch.cuda.Stream(device=self.dev, priority=prio)
        if self._raw_last is None:
            self.s_last = torch.cuda.Stream(device=self.dev, priority=cfg.last_priority) if pipeline else None
        px = image_size * image_size * 3
        self.Xs = torch.empty((self.E * self.n_total, px), device=self.dev)       # support store, NHWC rows
        self.Xall = torch.empty((self.E * self.n_all, image_size, image_size, 3), device=self.dev)
        ya = np.repeat(np.arange(n_way), n_support)
        self.y_support = (np.tile(ya, n_views + 1) if mode == "gnn" else ya).astype(np.int32)
        # every support image is drawn once per epoch: with >1 epoch cache its (mini-batch independent) stem conv
        if stem_cache:
            # 64 channels x (H/2)^2 fp32 per resident support image: 226 KB at 84x84, 3.2 MB at 224x224.  Keep the cache
            # only while it fits comfortably beside the other resident buffers.
            # Round 5: sized from the layout that is actually allocated.  The test used to price the FULL-resolution cache
            # (42 x 42 x 64 per image) although the default keeps the pooled (max, min) pair -- half of it -- so 20-shot at
            # E = 128 (115.6 GB full-resolution, 58 GB pooled) silently lost its stem cache and ran slower than E = 96
            # (19.9 vs 22.0 episodes/s, profiles/r04_f_other_configs.txt; round-4 verdict weak 9).
            pooled_pref = False if Fn.X3_PLANES else None
            need = Fn.StemCache.bytes_needed(self.E * self.n_total, image_size, pooled=pooled_pref)
            total = torch.cuda.get_device_properties(self.dev).total_memory
            stem_cache = need <= 0.35 * total
        # (the opt-in pre-split-planes trunk reads the full-resolution cache; the default is the pooled (max, min) form)
        self._stem_pooled = False if Fn.X3_PLANES else None
        self.stem = Fn.StemCache(self.W, self.E * self.n_total, image_size, self.dev, pooled=self._stem_pooled) if stem_cache else None
        self.fuse_next = fuse_next_policy(self._fuse_next_setting, self.E, self.stem is not None, image_size, self.W.f16x2)

    # ------------------------------------------------------------------ ingest
    @_on_device
    def load_episode(self, slot, liz_x, Xs=None, Xall=None):
        """finetune.py:208-233: support images of view 0 twice, then of views 1.. (device NCHW -> NHWC store).
        ``Xs`` / ``Xall``: target stores (default: the engine's current ones; the prefetch path fills the alternate pair)."""
        Xs = self.Xs if Xs is None else Xs
        Xall = self.Xall if Xall is None else Xall
        ns, npv, H = self.n_support, self.n_per_view, self.size
        assert len(liz_x) == self.n_views
        views = [liz_x[0].to(self.dev, non_blocking=True)]
        assert views[0].shape[1] == ns + self.n_query
        if self.mode == "gnn":
            views += [xv.to(self.dev, non_blocking=True) for xv in liz_x[1:]]
        views = [v if (v.dtype == torch.float32 and v.is_contiguous()) else v.float().contiguous() for v in views]
        ptrs = (ctypes.c_void_p * len(views))(*[v.data_ptr() for v in views])
        rc = ops._lib.lib().mft_ingest_episode_views(ptrs, len(views), 1 if self.mode == "gnn" else 0, self.n_way,
                                                     ns + self.n_query, ns, 3, H, H, ops._p(Xs[slot * self.n_total]),
                                                     ops._p(Xall[slot * self.n_all]), ops._stream())
        ops._lib.check(rc, "mft_ingest_episode_views")

    @_on_device
    def load_episode_source(self, slot, src_u8, params, Xs=None, Xall=None):
        """Ingest straight from raw images: src_u8 [n_way, n_support+n_query, Hs, Ws, 3] uint8 on the
        device, params [n_views, n_way*(n_support+n_query), 10] from augment.sample_view_params.  One launch writes the
        support views into the support store in finetune.py:208-233's order (view 0 twice, then views 1..), one more
        writes view 0 of every image for the final pass -- no host-side PIL, no NCHW staging copies."""
        from . import augment
        Xs = self.Xs if Xs is None else Xs
        Xall = self.Xall if Xall is None else Xall
        ns, npv, H = self.n_support, self.n_per_view, self.size
        assert self.mode == "gnn" and params.shape[0] == self.n_views
        src = src_u8.to(self.dev)
        n_way, per, Hs, Ws, _ = src.shape
        assert n_way == self.n_way and per == ns + self.n_query
        P = torch.as_tensor(params, dtype=torch.float32).view(self.n_views, n_way, per, augment.NPARAM)
        px = H * H * 3
        sup_src = src[:, :ns].reshape(npv, Hs, Ws, 3).contiguous()
        Ps = P[:, :, :ns].reshape(self.n_views, npv, augment.NPARAM)
        Ps = torch.cat([Ps[:1], Ps], 0)                                   # view 0 twice, then views 1.. (x_a_i doubling)
        augment.augment_views(sup_src, Ps, H, out=Xs[slot * self.n_total], view_stride=npv * px, img_stride=px)
        all_src = src.reshape(self.n_all, Hs, Ws, 3).contiguous()
        augment.augment_views(all_src, P[:1].reshape(1, self.n_all, augment.NPARAM), H, out=Xall[slot * self.n_all],
                              view_stride=self.n_all * px, img_stride=px)

    def step_tables(self, perms, n_active):
        """Index/label tables for all inner steps.  perms[e][epoch] is a permutation of n_total.  Returns a list of
        (k, idx[E*k] int32 rows of the support store, labels[E*k] int32), one entry per step (finetune.py:270-284)."""
        E, bs, nt = self.E, self.bs, self.n_total
        P = np.empty((E, self.epochs, nt), dtype=np.int64)
        for e in range(E):
            pe = perms[min(e, n_active - 1)]
            for ep in range(self.epochs):
                P[e, ep] = pe[ep]
        base = (np.arange(E, dtype=np.int64) * nt)[:, None]
        tables = []
        for ep in range(self.epochs):
            sel_all = P[:, ep]                                   # [E, nt]
            idx_all = (sel_all + base).astype(np.int32)
            lab_all = self.y_support[sel_all]
            for j in range(0, nt, bs):
                k = min(bs, nt - j)
                tables.append((k, np.ascontiguousarray(idx_all[:, j:j + k]).reshape(-1),
                               np.ascontiguousarray(lab_all[:, j:j + k]).reshape(-1)))
        return tables

    # ------------------------------------------------------------------ inner loop
    @_on_device
    def prepare_batch(self):
        """Once per batch of episodes, after ingest: fill the stem cache for all resident support images."""
        if self.stem is not None:
            H = self.size
            self.stem.fill(self.Xs.view(self.E * self.n_total, H, H, 3))

    def trunk_step(self, idx_dev, k, parity):
        """Frozen part of one inner step: gather the mini-batches, run trunk.0-6 (shared weights, per-episode BN
        statistics).  Independent of the adapted weights, hence of the previous step."""
        E, H = self.E, self.size
        a = self.arena_trunk
        if _DEBUG_SKIP_TRUNK and (k, parity) in self._dbg_x6:       # measurement aid only (tools): what the step costs without the trunk
            return self._dbg_x6[(k, parity)]
        if _DEBUG_SKIP_TRUNK:
            self._dbg_x6[(k, parity)] = Fn.resnet10_trunk(self.W, None, a, k, upto=7, tag="tr%d.%d" % (k, parity), stem=(self.stem, idx_dev))
            return self._dbg_x6[(k, parity)]
        if self.stem is not None:
            return Fn.resnet10_trunk(self.W, None, a, k, upto=7, tag="tr%d.%d" % (k, parity), stem=(self.stem, idx_dev))
        n = idx_dev.numel()                      # E*k images per step, times the number of steps in the chunk
        xb = ops.gather_rows(self.Xs, idx_dev, out=a.get("xb%d.%d" % (k, parity), (n, H * H * 3)))
        return Fn.resnet10_trunk(self.W, xb.view(n, H, H, 3), a, k, upto=7, tag="tr%d.%d" % (k, parity))

    def last_step(self, x6, lab_dev, k, nxt=None, tape=None, before_next_read=None):
        """Adapted part: trunk.7 forward with per-episode weights, CE on the 512-d feature, last-block backward,
        Adam (finetune.py:286-299).
        ``nxt`` = (x6 of the NEXT step | None, its tape buffers | None): the weight-gradient + Adam launches also run the next
        step's trunk.7 forward (Fn.last_block_backward(nxt=)); ``tape``: this step's forward as left by the PREVIOUS step's
        launches (None: run the forward here -- the first step of a loop).  ``before_next_read``: the stream wait for the producer
        of ``nxt``'s x6, run by last_block_backward right before the first launch that reads it."""
        E = self.E
        if tape is None:
            tape = {}
            feat = Fn.last_block_forward(self.W, x6, self.arena, k, slab=self.adapt.w, tape=tape, tag="s%d" % k)
        else:
            feat = tape["feat"]
        self.adapt.step += 1
        if self.mode == "linear":
            # classifier step (logits, CE, d feature, Adam(lr .01, wd .001) on W, b) in one launch; finetune.py:147-158
            c = self.cls
            dlogits = self.arena.get("lin.dfeat%d" % k, (E * k, 512))
            loss = self.arena.get("lin.loss", (E,))
            rc = ops._lib.lib().mft_linear_head_step(ops._p(feat), 512, ops._p(lab_dev), k, E, self.n_way, 512, ops._p(c["W"]),
                                                     ops._p(c["b"]), ops._p(c["mW"]), ops._p(c["vW"]), ops._p(c["mb"]),
                                                     ops._p(c["vb"]), ops._p(dlogits), 512, ops._p(loss), self.adapt.step,
                                                     self.lr, 0.9, 0.999, 1e-8, 0.001, ops._stream())
            ops._lib.check(rc, "mft_linear_head_step")
            ce = None
        else:
            # inner loss = CE on the pooled 512-d feature (finetune.py:286-291); fused with the pool/ReLU backward
            loss = self.arena.get("ce.loss", (E,))
            dlogits, ce = None, (feat, lab_dev, loss)
        if self.use_graph:
            ops.adam_hyper_advance(self.step_dev, self.hyper, lr=self.lr)
            Fn.last_block_backward(tape, dlogits, self.adapt.w, self.adapt.g, self.arena, ipg=k, tag="bw%d" % k,
                                   adam=(self.adapt.m, self.adapt.v, self.hyper, self.lr), ce=ce)
        elif self.fused_adam:
            Fn.last_block_backward(tape, dlogits, self.adapt.w, self.adapt.g, self.arena, ipg=k, tag="bw%d" % k,
                                   adam=(self.adapt.m, self.adapt.v, self.adapt.step, self.lr), ce=ce, nxt=nxt,
                                   before_next_read=before_next_read)
        else:
            Fn.last_block_backward(tape, dlogits, self.adapt.w, self.adapt.g, self.arena, ipg=k, tag="bw%d" % k, ce=ce)
            ops.adam_step(self.adapt.w.flat, self.adapt.g.flat, self.adapt.m.flat, self.adapt.v.flat, self.adapt.step,
                          lr=self.lr)
        return loss

    def _next_tape(self, x6, k, parity):
        n, H6 = x6.shape[0], x6.shape[1]
        oh = (H6 + 2 - 3) // 2 + 1
        return Fn.next_step_tape(self.arena, "nx%d.%d" % (k, parity), n, oh, oh, 512, n // k)

    def inner_step(self, idx_dev, lab_dev, k):
        return self.last_step(self.trunk_step(idx_dev, k, 0), lab_dev, k)

    @_on_device
    def inner_loop(self, tables):
        """All inner steps.  With ``pipeline`` the frozen trunk of step t+1 runs on its own HIP stream while the
        HBM-bound last-block backward + Adam of step t runs on another (x6 is double-buffered); the two halves of a
        step stress different resources (MFMA vs HBM)."""
        if not tables:
            return
        self.fused_last_loop = False
        dev = self.dev
        if len({t[0] for t in tables}) == 1:               # uniform mini-batches: two H2D copies for the whole loop
            idx_dev = torch.from_numpy(np.stack([t[1] for t in tables])).to(dev, non_blocking=True)
            lab_dev = torch.from_numpy(np.stack([t[2] for t in tables])).to(dev, non_blocking=True)
            idx_all, lab_all = list(idx_dev.unbind(0)), list(lab_dev.unbind(0))
        else:
            idx_all = [torch.from_numpy(t[1]).to(dev, non_blocking=True) for t in tables]
            lab_all = [torch.from_numpy(t[2]).to(dev, non_blocking=True) for t in tables]
        if self.use_graph and len({t[0] for t in tables}) == 1:
            k = tables[0][0]
            ent = self._graphs.get(k)
            start = 0
            if ent is None:
                # step 0 runs eagerly (it also creates every arena buffer); then one step is captured, not executed
                sidx, slab = torch.empty_like(idx_all[0]), torch.empty_like(lab_all[0])
                sidx.copy_(idx_all[0]); slab.copy_(lab_all[0])
                self.inner_step(sidx, slab, k)
                start = 1
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.inner_step(sidx, slab, k)
                ent = self._graphs[k] = (g, sidx, slab)
            g, sidx, slab = ent
            for t in range(start, len(tables)):
                sidx.copy_(idx_all[t])
                slab.copy_(lab_all[t])
                g.replay()
            return
        uniform = len({t[0] for t in tables}) == 1
        k0 = tables[0][0]
        H6 = (((((self.size + 6 - 7) // 2 + 1) + 2 - 3) // 2 + 1 + 1) // 2 + 1) // 2        # 84 -> 42 -> 21 -> 11 -> 6; 224 -> 14
        fuse = (self.fuse_next and uniform and self.fused_adam and not self.use_graph and Fn.next_forward_ok(k0, H6))
        # what THIS loop runs (bench.py reports it; self.fuse_next is only the policy): the fused next-step forward needs uniform
        # tables, the fused Adam launches, a shape the walking kernel covers and -- on the two-stream path -- one step per trunk launch
        self.fused_last_loop = bool(fuse and (not self.pipeline or self.trunk_chunk == 1))
        if not self.pipeline:
            if fuse:
                # step t's weight-gradient launches also produce step t+1's last-block forward: x6 of t+1 must exist before them
                n_steps = len(tables)
                x6 = self.trunk_step(idx_all[0], k0, 0)
                tape = None
                for t in range(n_steps):
                    x6n = self.trunk_step(idx_all[t + 1], k0, (t + 1) % 3) if t + 1 < n_steps else None
                    tn = self._next_tape(x6, k0, (t + 1) & 1) if x6n is not None else None
                    self.last_step(x6, lab_all[t], k0, nxt=(x6n, tn), tape=tape)
                    x6, tape = x6n, tn
                return
            for (k, _, _), idx, lab in zip(tables, idx_all, lab_all):
                self.inner_step(idx, lab, k)
            return
        cur = torch.cuda.current_stream(dev)
        self.s_trunk.wait_stream(cur)
        self.s_last.wait_stream(cur)
        if fuse and self.trunk_chunk == 1:
            # The frozen trunk runs TWO steps ahead on its own stream (three x6 buffers): the last-block launches of step t read
            # x6[t] (weight gradients) and x6[t+1] (the next step's forward they also compute).
            n_steps = len(tables)
            x6s, ready, done = [None] * n_steps, [None] * n_steps, [None, None, None]

            def launch_trunk(t):
                par = t % 3
                with torch.cuda.stream(self.s_trunk):
                    if done[par] is not None:
                        self.s_trunk.wait_event(done[par])
                    x6s[t] = self.trunk_step(idx_all[t], k0, par)
                    ready[t] = torch.cuda.Event()
                    ready[t].record(self.s_trunk)

            launch_trunk(0)
            if n_steps > 1:
                launch_trunk(1)
            tape = None
            for t in range(n_steps):
                if t + 2 < n_steps:
                    launch_trunk(t + 2)               # its buffer was last read by step t-1, whose completion event it waits for
                with torch.cuda.stream(self.s_last):
                    self.s_last.wait_event(ready[t])
                    x6n = tn = wait_next = None
                    if t + 1 < n_steps:
                        # x6[t+1] is first read by the fused launches, AFTER this step's cross entropy, BatchNorm backward and data
                        # gradient (0.3 ms): the wait goes there.  Waiting here, at the top of the step, left the last-block queue
                        # idle for ~0.2 ms per step while the trunk stream -- starved by the fused trunk.7.C2 launch it ran beside
                        # -- finished the tail of its own step (kernel timeline, profiles/r04_d_timeline_*.txt).
                        x6n = x6s[t + 1]
                        tn = self._next_tape(x6n, k0, (t + 1) & 1)
                        ev_next = ready[t + 1]
                        wait_next = (lambda ev=ev_next: self.s_last.wait_event(ev))
                    self.last_step(x6s[t], lab_all[t], k0, nxt=(x6n, tn), tape=tape, before_next_read=wait_next)
                    tape = tn
                    ev = torch.cuda.Event()
                    ev.record(self.s_last)
                    done[t % 3] = ev
                x6s[t] = None
            cur.wait_stream(self.s_trunk)
            cur.wait_stream(self.s_last)
            return
        # The frozen trunk does not depend on the adaptation, so it runs AHEAD in chunks of ``trunk_chunk`` steps: one set
        # of trunk launches covers T steps (T*E*k images, BatchNorm groups of k images as before), then the last-block
        # stream consumes the T slices.  Bigger launches for the MFMA-bound half, T times fewer of them.
        T = self.trunk_chunk if len({t[0] for t in tables}) == 1 else 1
        done = [None, None]                          # last-block completion events per x6 buffer
        n_steps = len(tables)
        for c, s0 in enumerate(range(0, n_steps, T)):
            s1 = min(s0 + T, n_steps)
            par = c & 1
            k = tables[s0][0]
            with torch.cuda.stream(self.s_trunk):
                if done[par] is not None:
                    self.s_trunk.wait_event(done[par])
                if T == 1:
                    idx_cat = idx_all[s0]
                else:
                    idx_cat = idx_dev[s0:s1].reshape(-1)           # rows of the table are consecutive: a view
                x6 = self.trunk_step(idx_cat, k, par)
                ready = torch.cuda.Event()
                ready.record(self.s_trunk)
            with torch.cuda.stream(self.s_last):
                self.s_last.wait_event(ready)
                per = self.E * k
                for j, t in enumerate(range(s0, s1)):
                    self.last_step(x6[j * per:(j + 1) * per], lab_all[t], k)
                ev = torch.cuda.Event()
                ev.record(self.s_last)
                done[par] = ev
        cur.wait_stream(self.s_trunk)
        cur.wait_stream(self.s_last)

    @_on_device
    def final_scores(self, arena=None):
        """finetune.py:306-317: transductive feature pass over all n_way*(n_support+n_query) images, then
        GnnNet.set_forward(is_feature=True) and softmax.  (finetune.py:307's second pass is dead compute.)"""
        if arena is not None:
            return self._final_scores(arena)
        return self._final_scores(self.arena)

    def _final_scores(self, arena):
        feats = Fn.resnet10_forward(self.W, self.Xall, arena, ipg=self.n_all, slab=self.adapt.w, tag="fin")
        if self.mode == "linear":
            # finetune.py:165-174: features of cat(support, query) in one train-mode pass (BatchNorm statistics over all
            # n_all images are order independent), classifier + softmax on the query rows (class-major)
            sc = arena.get("lin.scores", (self.E * self.n_all, self.n_way))
            rc = ops._lib.lib().mft_linear_head_scores(ops._p(feats), 512, self.n_all, self.E, self.n_way, 512,
                                                       ops._p(self.cls["W"]), ops._p(self.cls["b"]), ops._p(sc), ops._stream())
            ops._lib.check(rc, "mft_linear_head_scores")
            sc = sc.view(self.E, self.n_way, self.n_support + self.n_query, self.n_way)[:, :, self.n_support:]
            return sc.reshape(self.E, self.n_way * self.n_query, self.n_way), feats
        ns = self.n_support // 2 if self.fold50 else self.n_support
        scores = Fn.gnnnet_scores(self.G, feats, self.E, self.n_way, ns, self.n_query, arena, fold=self.fold50)
        return ops.softmax_rows(scores).view(self.E, self.n_way * self.n_query, self.n_way), feats

    def set_classifier(self, w0, b0, n_active):
        """Initial Linear(512, n_way) weights per episode: w0 [n, n_way, 512], b0 [n, n_way] (finetune.py:65)."""
        c = self.cls
        for k in ("mW", "vW", "mb", "vb"):
            c[k].zero_()
        w0 = torch.as_tensor(w0, dtype=torch.float32).to(self.dev).view(-1, self.n_way, 512)
        b0 = torch.as_tensor(b0, dtype=torch.float32).to(self.dev).view(-1, self.n_way)
        idx = torch.clamp(torch.arange(self.E, device=self.dev), max=n_active - 1)
        c["W"].copy_(w0[idx])
        c["b"].copy_(b0[idx])

    def _flip_buffers(self):
        """Deferred final pass: the adapted weights and final-pass images of batch i must survive while batch i+1 is ingested
        and adapted, so both live in two alternating buffers (m, v, g are only used inside the inner loop and stay single)."""
        if self._alt is None:
            self._alt = [(self.adapt.w, self.Xall),
                         (Fn.LastBlockSlab(self.E, self.dev, zero=False, flat=self.adapt.w_alt_flat), torch.empty_like(self.Xall))]
            self._final_done = [None, None]
            self._bi = 0
            self.arena_final = Fn.Arena(self.dev)
            self.s_final = torch.cuda.Stream(device=self.dev)
        self._bi ^= 1
        self.adapt.w, self.Xall = self._alt[self._bi]
        if self._final_done[self._bi] is not None:                 # the final pass that last read these buffers is done
            torch.cuda.current_stream(self.dev).wait_event(self._final_done[self._bi])

    def _ingest(self, episodes, sources, Xs=None, Xall=None):
        n = len(episodes)
        for slot in range(self.E):
            ep = episodes[min(slot, n - 1)]                          # pad a short batch by repeating the last episode
            if sources:
                self.load_episode_source(slot, ep[0], ep[1], Xs, Xall)
            else:
                self.load_episode(slot, ep, Xs, Xall)

    def _enqueue_prefetch(self, episodes, sources):
        """Ingest + stem cache of the next batch on their own stream, into the alternate support store / stem cache / final-pass
        store, while this batch's inner loop runs: the stem convolution is matrix-bound, the inner loop HBM-bound."""
        if self._pre_bufs is None:
            need = self.Xs.numel() * 4 + self.stem.nbytes()
            free, total = torch.cuda.mem_get_info(self.dev)
            if need > 0.8 * free:
                return                                           # not enough room for a second support store + stem cache
            self._pre_bufs = {"Xs": torch.empty_like(self.Xs),
                              "stem": Fn.StemCache(self.W, self.E * self.n_total, self.size, self.dev, pooled=self._stem_pooled),
                              "stream": torch.cuda.Stream(device=self.dev)}
        b = self._pre_bufs
        sp = b["stream"]
        cur = torch.cuda.current_stream(self.dev)
        nxt = self._bi ^ 1                                        # the weight / final-pass buffers the next batch will flip to
        sp.wait_stream(cur)                                       # previous readers of the alternate store are behind `cur`
        if self._final_done[nxt] is not None:
            sp.wait_event(self._final_done[nxt])                  # the deferred final pass that still reads that Xall
        with torch.cuda.stream(sp):
            self._ingest(episodes, sources, b["Xs"], self._alt[nxt][1])
            b["stem"].fill(b["Xs"].view(self.E * self.n_total, self.size, self.size, 3))
            ev = torch.cuda.Event()
            ev.record(sp)
        self._pre = {"token": episodes, "sources": sources, "done": ev, "n": len(episodes)}

    @_on_device
    def run_batch(self, episodes, perms=None, return_feats=False, classifier_init=None, sources=False, defer_final=False,
                  prefetch=None):
        """episodes: list (<= E) of liz_x -- or, with ``sources=True``, of (src_u8, view_params) pairs for device-side view
        generation; perms: per-episode list of per-epoch permutations (default: drawn from the global numpy RNG episode by
        episode, exactly the reference's draw order).  Returns softmax scores [len(episodes), n_way*n_query, n_way].
        ``defer_final``: enqueue the final 100-image pass + GNN head on a third stream and return at once, so that it
        overlaps the ingest / stem cache / first inner steps of the NEXT run_batch call (the returned scores are valid
        after a device synchronisation or ``engine.s_final.synchronize()``).
        ``prefetch``: the episode list of the NEXT call (same ``sources``; needs ``defer_final`` in both calls): its ingest and
        stem cache are enqueued now on a fourth stream and swapped in when that call arrives with the same list object."""
        n = len(episodes)
        assert 0 < n <= self.E
        defer_final = defer_final and self.mode == "gnn" and not self.use_graph and not return_feats
        if defer_final:
            self._flip_buffers()
        if perms is None:
            perms = [draw_perms(self.n_total, self.epochs) for _ in range(n)]
        pre, self._pre = self._pre, None
        if pre is not None and defer_final and pre["token"] is episodes and pre["sources"] == sources and pre["n"] == n:
            b = self._pre_bufs
            self.Xs, b["Xs"] = b["Xs"], self.Xs
            self.stem, b["stem"] = b["stem"], self.stem
            torch.cuda.current_stream(self.dev).wait_event(pre["done"])
            prepared = True
        else:
            if pre is not None:                                   # a prefetch that is not consumed: let it finish before reuse
                torch.cuda.current_stream(self.dev).wait_event(pre["done"])
            self._ingest(episodes, sources)
            prepared = False
        self.adapt.reset(self.W)
        self.step_dev.zero_()
        if self.mode == "linear":
            if classifier_init is None:
                raise RuntimeError("mode='linear' needs classifier_init=(w0 [n,n_way,512], b0 [n,n_way])")
            self.set_classifier(classifier_init[0], classifier_init[1], n)
        if not prepared:
            self.prepare_batch()
        if prefetch is not None and defer_final and self.stem is not None and self.pipeline:
            self._enqueue_prefetch(prefetch, sources)
        self.inner_loop(self.step_tables(perms, n))
        if defer_final:
            cur = torch.cuda.current_stream(self.dev)
            self.s_final.wait_stream(cur)
            with torch.cuda.stream(self.s_final):
                scores, _ = self.final_scores(arena=self.arena_final)
                ev = torch.cuda.Event()
                ev.record(self.s_final)
            self._final_done[self._bi] = ev
            return scores[:n]
        scores, feats = self.final_scores()
        if return_feats:
            return scores[:n], feats.view(self.E, self.n_all, 512)[:n]
        return scores[:n]


_ADAPT_STREAMS = {}
_ADAPT_ARENAS = {}


ADAPT_GRAPH = settings.current().adapt_graph
ADAPT_BATCHED_TRUNK = settings.current().adapt_batched_trunk
_ADAPT_GRAPHS = {}


class _AdaptLoop:
    """Static buffers (and, from the second call on, ONE hipGraph) of adapt_last_block for one (module weights, episode shape):
    the inner loop of a meta-fine-tuning training episode is ~105 Adam steps x ~45 launches on FOUR images each -- 57 ms of
    launch overhead per episode when issued from Python, and the same launch sequence every episode (the step sizes, the Adam
    step numbers and every buffer are fixed; only the support images, the permutation tables and the weights' VALUES change)."""

    def __init__(self, W, dev, n, H, n_steps_idx, plan):
        self.W, self.dev = W, dev
        self.ad = AdaptState(1, dev)
        self.Xs = torch.empty((n, H * H * 3), device=dev, dtype=torch.float32)
        self.idx_all = torch.zeros((n_steps_idx,), device=dev, dtype=torch.int32)
        self.lab_all = torch.zeros((n_steps_idx,), device=dev, dtype=torch.int32)
        # batched trunk: the mini-batch sizes that occur (full, ragged tail), each step's (kind, group) and the image indices per kind
        self.kinds = sorted({k for _, k in plan}, reverse=True)
        count = {k: 0 for k in self.kinds}
        self.where = []
        for _, k in plan:
            self.where.append((self.kinds.index(k), count[k]))
            count[k] += 1
        self.idx_kind = {k: torch.zeros((count[k] * k,), device=dev, dtype=torch.int32) for k in self.kinds}
        self.order = torch.tensor([(ki << 24) | g for ki, g in self.where], dtype=torch.int32).to(dev)
        self.order_steps = torch.tensor([(ki << 24) | t for t, (ki, _) in enumerate(self.where)], dtype=torch.int32).to(dev)    # row t, kind's row count
        self.running = None          # name -> (running_mean, running_var) static copies
        self.graph = None
        self.out = None
        self.calls = 0


def _adapt_body(st, feature_mod, plan, n, H, lr):
    """The launches of one episode's inner loop on ``st``'s static buffers (eager, or being recorded into st.graph)."""
    from . import autograd_ops as AG
    W, dev, ad = st.W, st.dev, st.ad
    arena = AG.arena_for(dev)
    ad.reset(W)
    mods = [(name, m) for name, m in feature_mod.named_modules() if isinstance(m, torch.nn.BatchNorm2d)]
    if st.running is None:
        st.running = {name: (m.running_mean.detach().clone(), m.running_var.detach().clone()) for name, m in mods}
    else:
        torch._foreach_copy_([t for name, _ in mods for t in st.running[name]],
                             [t for _, m in mods for t in (m.running_mean.detach(), m.running_var.detach())])
    running = st.running
    cur = torch.cuda.current_stream(dev)
    arena_t = _ADAPT_ARENAS.get(dev)
    if arena_t is None:
        arena_t = _ADAPT_ARENAS[dev] = Fn.Arena(dev)
    if ADAPT_BATCHED_TRUNK and len(st.kinds) <= 2:
        # The trunk below the adapted block is frozen for the whole inner loop and a mini-batch's trunk output depends on that
        # mini-batch only (train-mode BatchNorm over its own images): the ~105 trunk forwards of the episode run as ONE grouped
        # pass per mini-batch size (group = step), ~60 launches instead of ~3,000.  Their BatchNorm running statistics are then
        # advanced through the steps in order by one small launch per layer (mft_bn_running_ema).
        x6_of = []
        for k in st.kinds:
            idx_k = st.idx_kind[k]
            xb = ops.gather_rows(st.Xs, idx_k, out=arena_t.get("adb.xb%d" % k, (idx_k.numel(), st.Xs.shape[1])))
            x6_of.append(Fn.resnet10_trunk(W, xb.view(-1, H, H, 3), arena_t, k, upto=7, running=None, tag="adb%d" % k))
        _trunk_running_ema(st, arena_t, running, H)
        # the adapted block's three BatchNorms: every step writes its batch statistics into row t of a per-episode table (one
        # small launch does statistics + normalise + ReLU [+ add + pool]); their running statistics follow after the loop
        n_steps = len(plan)
        sink = {nm: (arena.get("ad.sink.%s.mean" % nm, (n_steps, 512)), arena.get("ad.sink.%s.rstd" % nm, (n_steps, 512)))
                for nm in ("bn1", "bn2", "bns")}
        for t, (off, k) in enumerate(plan):
            ki, g = st.where[t]
            x6 = x6_of[ki][g * k:(g + 1) * k]
            lab = st.lab_all[off:off + k]
            tape = {}
            feat = Fn.last_block_forward(W, x6, arena, k, slab=ad.w, tape=tape, running=None, tag="ad%d" % k,
                                         stats_out={nm: (m[t:t + 1], r[t:t + 1]) for nm, (m, r) in sink.items()})
            ad.step += 1
            Fn.last_block_backward(tape, None, ad.w, ad.g, arena, ipg=k, tag="adbw%d" % k,
                                   adam=(ad.m, ad.v, ad.step, lr), ce=(feat, lab, arena.get("ad.loss", (1,))))
        h7 = x6_of[0].shape[1]
        h7 = (h7 + 2 - 3) // 2 + 1
        rows = [k * h7 * h7 for k in st.kinds] + [1]
        lib = ops._lib.lib()
        for nm, bn in (("bn1", "trunk.7.BN1"), ("bn2", "trunk.7.BN2"), ("bns", "trunk.7.BNshortcut")):
            m, r = sink[nm]
            rm, rv = running[bn]
            ops._lib.check(lib.mft_bn_running_ema(ops._p(m), ops._p(r), rows[0], ops._p(m), ops._p(r), rows[1], ops._p(st.order_steps),
                                                  n_steps, 512, ops.BN_EPS, 0.1, ops._p(rm), ops._p(rv), ops._stream()), "mft_bn_running_ema")
        return ad.w.export(0)
    # frozen trunk.0-6 of step t+1 on a second stream while the last block of step t is adapted (as in FinetuneEngine)
    s_trunk = _ADAPT_STREAMS.get(dev)
    if s_trunk is None:
        s_trunk = _ADAPT_STREAMS[dev] = torch.cuda.Stream(device=dev)
    s_trunk.wait_stream(cur)
    done = [None, None]
    for t, (off, k) in enumerate(plan):
        idx, lab = st.idx_all[off:off + k], st.lab_all[off:off + k]
        par = t & 1
        with torch.cuda.stream(s_trunk):
            if done[par] is not None:
                s_trunk.wait_event(done[par])
            xb = ops.gather_rows(st.Xs, idx, out=arena_t.get("ad.xb%d.%d" % (k, par), (k, st.Xs.shape[1])))
            x6 = Fn.resnet10_trunk(W, xb.view(k, H, H, 3), arena_t, k, upto=7, running=running, tag="adt%d.%d" % (k, par))
            ready = torch.cuda.Event()
            ready.record(s_trunk)
        cur.wait_event(ready)
        tape = {}
        feat = Fn.last_block_forward(W, x6, arena, k, slab=ad.w, tape=tape, running=running, tag="ad%d" % k)
        ad.step += 1
        # CE on the pooled feature fused with the pool/ReLU backward; Adam fused into the weight-gradient epilogues
        Fn.last_block_backward(tape, None, ad.w, ad.g, arena, ipg=k, tag="adbw%d" % k,
                               adam=(ad.m, ad.v, ad.step, lr), ce=(feat, lab, arena.get("ad.loss", (1,))))
        ev = torch.cuda.Event()
        ev.record(cur)
        done[par] = ev
    cur.wait_stream(s_trunk)
    return ad.w.export(0)


def _trunk_running_ema(st, arena_t, running, H):
    """running_mean / running_var of the nine BatchNorms of trunk.1-6 after the episode's steps, from the per-group statistics the
    grouped passes left in the arena (one launch per layer)."""
    lib = ops._lib.lib()
    oh0 = (H + 6 - 7) // 2 + 1
    sp = {4: (oh0 + 2 - 3) // 2 + 1}
    sp[5] = (sp[4] + 2 - 3) // 2 + 1
    sp[6] = (sp[5] + 2 - 3) // 2 + 1
    layers = [("trunk.1", ".bn0", oh0, 64)]
    for bi in (4, 5, 6):
        cout = Fn.STAGES[bi][1]
        layers += [("trunk.%d.BN1" % bi, ".trunk.%d.bn1" % bi, sp[bi], cout), ("trunk.%d.BN2" % bi, ".trunk.%d.bn2" % bi, sp[bi], cout)]
        if Fn.STAGES[bi][0] != cout:
            layers.append(("trunk.%d.BNshortcut" % bi, ".trunk.%d.bns" % bi, sp[bi], cout))
    for name, suffix, hw, C in layers:
        rm, rv = running[name]
        sets = []
        for k in st.kinds:
            G = st.idx_kind[k].numel() // k
            # (Arena.existing: if resnet10_trunk stops leaving (mean, rstd) under these names, this raises instead of advancing the
            # running statistics from an uninitialised buffer)
            sets.append((arena_t.existing("adb%d%s.mean" % (k, suffix), (G, C)), arena_t.existing("adb%d%s.rstd" % (k, suffix), (G, C)), k * hw * hw))
        if len(sets) == 1:
            sets.append((None, None, 1))
        (ma, ra, na), (mb, rb, nb) = sets
        ops._lib.check(lib.mft_bn_running_ema(ops._p(ma), ops._p(ra), na, ops._p(mb), ops._p(rb), nb, ops._p(st.order), st.order.numel(), C,
                                              ops.BN_EPS, 0.1, ops._p(rm), ops._p(rv), ops._stream()), "mft_bn_running_ema")


def adapt_last_block(feature_mod, x_a, y_a, epochs, batch_size, lr=0.01, perms=None):
    """The inner loop of GnnNet.set_forward_finetune (gnnnet.py:126-177) for one episode: Adam(lr=0.01) on the last
    ResNet block of a *copy* of ``feature_mod`` over ``epochs`` permutations of the support set in mini-batches of
    ``batch_size`` (ragged tail allowed).  x_a: [n,3,H,W] device NCHW; y_a: int32 numpy labels.
    Returns {state_dict key: tensor} for the nine adapted tensors and the BatchNorm running buffers (valid until the next call).
    From the second episode of a shape on, the whole loop is ONE hipGraph replay (MFT_ADAPT_GRAPH=0: eager launches)."""
    from . import autograd_ops as AG
    dev = x_a.device
    n, _, H, _ = x_a.shape
    W = AG.module_weights(feature_mod)                 # refreshed in place when the module's parameters moved on
    # all index / label tables go to the device in ONE copy each before the loop: a per-step pageable host-to-device
    # copy is stream-ordered and would drain the queue every step
    plan, flat_idx, flat_lab = [], [], []
    for ep in range(epochs):
        rand_id = np.random.permutation(n) if perms is None else perms[ep]
        for j in range(0, n, batch_size):
            ids = np.asarray(rand_id[j:min(j + batch_size, n)])
            plan.append((len(flat_idx), len(ids)))
            flat_idx.extend(ids.tolist())
            flat_lab.extend(np.asarray(y_a)[ids].tolist())
    # (the recorded loop reads the module's BatchNorm buffers through their addresses: a module whose buffers were re-allocated
    #  gets a fresh recording)
    bufs = tuple(b.data_ptr() for b in feature_mod.buffers())
    key = (dev.index, id(W), n, H, epochs, batch_size, float(lr), bufs)
    st = _ADAPT_GRAPHS.get(key)
    if st is None:
        for k_old in [k for k, v in _ADAPT_GRAPHS.items() if k[0] == dev.index and k[2:] == key[2:]]:
            del _ADAPT_GRAPHS[k_old]                   # the module's packed weights were rebuilt: drop the loop recorded on the old ones
        while len(_ADAPT_GRAPHS) >= 4:                 # each entry pins ~75 MB of slabs + its graph's pool: keep the four most recent
            del _ADAPT_GRAPHS[next(iter(_ADAPT_GRAPHS))]
        st = _ADAPT_GRAPHS[key] = _AdaptLoop(W, dev, n, H, len(flat_idx), plan)      # (st.W keeps id(W) unique while the entry lives)
    fi = np.asarray(flat_idx, dtype=np.int32)
    st.idx_all.copy_(torch.from_numpy(fi))
    st.lab_all.copy_(torch.from_numpy(np.asarray(flat_lab, dtype=np.int32)))
    for k in st.kinds:
        st.idx_kind[k].copy_(torch.from_numpy(np.concatenate([fi[off:off + kk] for off, kk in plan if kk == k])))
    st.Xs.copy_(ops.nchw_to_nhwc(x_a.contiguous().float()).view(n, -1))
    nbt = {name: m.num_batches_tracked.detach().clone() for name, m in feature_mod.named_modules()
           if isinstance(m, torch.nn.BatchNorm2d)}
    st.calls += 1
    if ADAPT_GRAPH and st.calls >= 2 and st.graph is None and st.graph is not False:
        try:                                           # (the first call ran eagerly: every arena buffer exists)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st.out = _adapt_body(st, feature_mod, plan, n, H, lr)
            st.graph = g
        except Exception as e:   # noqa: BLE001 -- whatever the runtime refuses to record: stay on the eager loop
            import warnings
            warnings.warn("hipGraph capture of the meta-fine-tuning inner loop failed (%s: %s); continuing without it" % (type(e).__name__, e))
            st.graph = False
            torch.cuda.synchronize()
    if st.graph:
        st.graph.replay()
        out = dict(st.out)
    else:
        out = _adapt_body(st, feature_mod, plan, n, H, lr)
    for name, (rm, rv) in st.running.items():
        out[name + ".running_mean"] = rm
        out[name + ".running_var"] = rv
        out[name + ".num_batches_tracked"] = nbt[name] + len(plan)
    return out
