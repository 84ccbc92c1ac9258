"""One typed configuration object for every knob of the package (round-4 verdict "weak 11": 43 scattered ``os.environ`` reads).

``Settings`` is a frozen dataclass: one field per knob, with its type, default, environment variable and meaning in ONE table
(``KNOBS``).  ``current()`` returns the Settings for the present environment -- parsed and validated once per distinct set of
``MFT_*`` values (tests that change a variable get a fresh object, everything else gets the cached one) -- but note
``IMPORT_TIME``: the knobs listed there are frozen into module constants at first import, the rest are read where they are used.
A value that does not parse raises a ValueError naming its variable; boolean knobs take 0 / 1 / true / false / yes / no / on / off.  No module of the
package parses an ``MFT_*`` variable itself any more; process-rank variables set by the launcher (RANK, LOCAL_RANK, WORLD_SIZE,
MASTER_ADDR, OMP_NUM_THREADS) and the build's HIPCC are not knobs and stay where they are used.

Defaults ARE the product; every non-default value is an A/B or test hook and says so."""
import dataclasses
import os


def _flag(v):
    lv = v.strip().lower()
    if lv in ("1", "true", "yes", "on"):
        return True
    if lv in ("0", "false", "no", "off"):
        return False
    raise ValueError("expected 0 / 1 (or true / false), got %r" % (v,))


def _opt_int(v):
    return None if v in ("", "none", "None") else int(v)


def _fuse(v):
    if v not in ("auto", "0", "1"):
        raise ValueError("expected auto, 0 or 1, got %r" % (v,))
    return v


# field, environment variable, parser, default, meaning
KNOBS = (
    # ---- engine (test-time fine-tune, engine.py)
    ("fuse_next", "MFT_FUSE_NEXT", _fuse, "auto", "fused weight-gradient + Adam + next-step forward launches: auto = engine.fuse_next_policy | 1 | 0"),
    ("wf_xcd", "MFT_WF_XCD", _flag, True, "fused launch: all workgroups of an episode on one XCD (A/B hook; read once per process)"),
    ("slab_candidates", "MFT_SLAB_CANDIDATES", _opt_int, None, "candidate buffers of the w/m/v placement scan (None: engine default 12; 0: off)"),
    ("slab_ballast_gb", "MFT_SLAB_BALLAST_GB", float, 12.0, "throw-away allocation between groups of placement candidates, GB"),
    ("slab_hints", "MFT_SLAB_HINTS", _flag, True, "re-use an earlier scan's placement when one probe confirms it"),
    ("trunk_chunk", "MFT_TRUNK_CHUNK", _opt_int, None, "inner steps per frozen-trunk launch set (None: 32 at E = 1, else 1)"),
    ("trunk_priority", "MFT_TRUNK_PRIORITY", int, 1, "HIP priority of the frozen-trunk stream (1 = the device's least)"),
    ("last_priority", "MFT_LAST_PRIORITY", int, -1, "torch priority of the last-block stream"),
    ("trunk_cus_per_xcd", "MFT_TRUNK_CUS_PER_XCD", int, 0, "CU-masked streams: CUs per XCD for the trunk stream (0 = no partition; measured slower)"),
    ("trunk_cus_only", "MFT_TRUNK_CUS_ONLY", _flag, False, "with trunk_cus_per_xcd: confine only the trunk stream"),
    ("debug_skip_trunk", "MFT_DEBUG_SKIP_TRUNK", _flag, False, "measurement aid: reuse one cached trunk output (results are WRONG)"),
    ("adapt_graph", "MFT_ADAPT_GRAPH", _flag, True, "adapt_last_block: replay the meta-fine-tuning inner loop from one hipGraph"),
    ("adapt_batched_trunk", "MFT_ADAPT_BATCHED_TRUNK", _flag, True, "adapt_last_block: frozen trunk of all steps as two grouped passes"),
    # ---- kernel-path selection (functional.py)
    ("fused_dgrad", "MFT_FUSED_DGRAD", _flag, False, "one-pass trunk.7.C2 weight gradient + Adam + data gradient (measured slower in the step)"),
    ("x3_planes", "MFT_X3_PLANES", _flag, False, "trunk activations pre-split into bf16x3 planes (measured slower)"),
    ("x3_fused_stats", "MFT_X3_FUSED_STATS", _flag, True, "BatchNorm statistics from the split-precision convolution epilogue"),
    ("x3_fold_bn", "MFT_X3_FOLD_BN", _flag, True, "frozen blocks without helper launches (per-tile partials merged by the consumers)"),
    ("trunk_f16x2", "MFT_TRUNK_F16X2", _flag, True, "frozen trunk.4-6 on two fp16 pieces per operand (0: three bf16 pieces)"),
    ("fuse_next_c2_only", "MFT_FUSE_NEXT_C2_ONLY", _flag, False, "only trunk.7.C2's launch fused (measured slower)"),
    ("stem_pooled", "MFT_STEM_POOLED", _flag, True, "stem cache keeps per-window (max, min) instead of the full-resolution output"),
    ("stem_fused_fill", "MFT_STEM_FUSED_FILL", _flag, True, "stem cache filled by ONE launch (convolution + moments + window min/max; 0: three launches through a full-resolution buffer)"),
    ("stem_chunk", "MFT_STEM_CHUNK", _opt_int, None, "images per stem-cache fill launch (None: StemCache's default)"),
    ("fused_last_block", "MFT_FUSED_LAST_BLOCK", _flag, True, "conv + BatchNorm fusions of the adapted block (csrc/skinny.hip)"),
    ("small_groups", "MFT_SMALL_GROUPS", int, 12, "up to this many per-episode weight sets take the K-sliced GEMM route"),
    ("fused_pair_mlp", "MFT_FUSED_PAIR_MLP", _flag, True, "Wcompute on the fused upper-triangle pair-MLP kernels"),
    ("pair_mlp_gb", "MFT_PAIR_MLP_GB", float, 6.0, "memory budget of the fused pair-MLP's hidden activations per chunk, GB"),
    # ---- training path
    ("train_graph", "MFT_TRAIN_GRAPH", _flag, True, "MetaTemplate episode loop: loss + backward replayed from one hipGraph"),
    ("train_x3", "MFT_TRAIN_X3", _flag, True, "meta-training: 3x3 layers with >= 8192 output rows on the bf16x3 kernels, forward and stride-1 data gradient (0: fp32 MFMA everywhere)"),
    ("train_x3_min_rows", "MFT_TRAIN_X3_MIN_ROWS", int, 6000, "meta-training: output rows from which a 3x3 layer takes the bf16x3 kernels (below: fp32 MFMA with K slices; measured at 3,780 / 945 rows in profiles/r06_q_train_x3_min_rows.txt)"),
    ("wgrad_batch", "MFT_WGRAD_BATCH", _flag, True, "meta-training backward: every layer's weight gradient deferred to the end of the pass and run in one multi-problem launch pair per 16 layers (0: one launch pair per layer, as round 5; bit-identical)"),
    ("pair_f16x2", "MFT_PAIR_F16X2", _flag, False, "GNN pair-MLP layers (Wcompute) as f16x2 products on the fp16 matrix cores (fp32-accurate; 0: fp32 MFMA)"),
    ("gemm_rk_rows", "MFT_GEMM_RK_ROWS", int, 4096, "meta-training: head linear layers (fc, Gconv.fc) of at most this many rows take the skinny register-K GEMM (0: the tile kernel always)"),
    ("pair_rk_rows", "MFT_PAIR_RK_ROWS", int, 16384, "meta-training: Wcompute layers over at most this many pair rows (all episodes of the step) take the register-K small-problem kernel, 32-row tiles, no LDS staging (0: the 128-row tile kernel always)"),
    ("train_source", "MFT_TRAIN_SOURCE", str, "pool", "train.main --dataset miniImageNet: 'pool' = resident uint8 class pool, 'synthetic' = host fp32 episodes"),
    # ---- drivers (finetune.main / train.main)
    ("standin_weights", "MFT_STANDIN_WEIGHTS", _flag, False, "allow synthetic stand-in weights when no checkpoint is found (explicit opt-in)"),
    ("image_size", "MFT_IMAGE_SIZE", int, 84, "image side the CLI evaluates at (84 = BASELINE; 224 = the reference's)"),
    ("episodes", "MFT_EPISODES", _opt_int, None, "override of the CLI's episode count (None: 600)"),
    ("episodes_per_batch", "MFT_EPISODES_PER_BATCH", _opt_int, None, "override of the CLI's lockstep batch (None: 128 / 96 / 64 by n_shot)"),
    ("timings", "MFT_TIMINGS", _flag, False, "print the drivers' phase timings to stderr"),
    ("pool_per_class", "MFT_POOL_PER_CLASS", _opt_int, None, "images per class of the dataset-shaped resident pool (None: the dataset's)"),
    ("synth_on_host", "MFT_SYNTH_ON_HOST", _flag, False, "generate synthetic test episodes with numpy on the host instead of on the device"),
    ("balance_batches", "MFT_BALANCE_BATCHES", _flag, True, "equalise the lockstep batch sizes of an evaluation"),
    # ---- distributed / test hooks
    ("force_collectives", "MFT_FORCE_COLLECTIVES", _flag, False, "run the N-rank collectives even with one rank (1-GPU RCCL tests)"),
    ("one_device", "MFT_ONE_DEVICE", _flag, False, "test hook: W ranks share device 0, gloo for the gathers"),
)


@dataclasses.dataclass(frozen=True)
class Settings:
    fuse_next: str = "auto"
    wf_xcd: bool = True
    slab_candidates: "int | None" = None
    slab_ballast_gb: float = 12.0
    slab_hints: bool = True
    trunk_chunk: "int | None" = None
    trunk_priority: int = 1
    last_priority: int = -1
    trunk_cus_per_xcd: int = 0
    trunk_cus_only: bool = False
    debug_skip_trunk: bool = False
    adapt_graph: bool = True
    adapt_batched_trunk: bool = True
    fused_dgrad: bool = False
    x3_planes: bool = False
    x3_fused_stats: bool = True
    x3_fold_bn: bool = True
    trunk_f16x2: bool = True
    fuse_next_c2_only: bool = False
    stem_pooled: bool = True
    stem_fused_fill: bool = True
    stem_chunk: "int | None" = None
    fused_last_block: bool = True
    small_groups: int = 12
    fused_pair_mlp: bool = True
    pair_mlp_gb: float = 6.0
    train_graph: bool = True
    train_x3: bool = True
    train_x3_min_rows: int = 6000
    wgrad_batch: bool = True
    pair_f16x2: bool = False
    gemm_rk_rows: int = 4096
    pair_rk_rows: int = 16384
    train_source: str = "pool"
    standin_weights: bool = False
    image_size: int = 84
    episodes: "int | None" = None
    episodes_per_batch: "int | None" = None
    timings: bool = False
    pool_per_class: "int | None" = None
    synth_on_host: bool = False
    balance_batches: bool = True
    force_collectives: bool = False
    one_device: bool = False

    @classmethod
    def from_env(cls, env=None):
        env = os.environ if env is None else env
        kw = {}
        for field, var, parse, default, _ in KNOBS:
            raw = env.get(var)
            if raw is None:
                kw[field] = default
                continue
            try:
                kw[field] = parse(raw)
            except (TypeError, ValueError) as e:
                raise ValueError("%s=%r: %s" % (var, raw, e)) from None
        s = cls(**kw)
        if s.train_source not in ("pool", "synthetic"):
            raise ValueError("MFT_TRAIN_SOURCE=%r: expected pool or synthetic" % (s.train_source,))
        if s.image_size < 32 or not 0 <= s.trunk_cus_per_xcd < 32:
            raise ValueError("MFT_IMAGE_SIZE / MFT_TRUNK_CUS_PER_XCD out of range")
        return s


assert tuple(f.name for f in dataclasses.fields(Settings)) == tuple(k[0] for k in KNOBS), "KNOBS and Settings fields must match"
_VARS = tuple(k[1] for k in KNOBS)
_cache = (None, None)


def current():
    """The Settings of the present environment (cached per distinct tuple of MFT_* values)."""
    global _cache
    key = tuple(os.environ.get(v) for v in _VARS)
    if _cache[0] != key:
        _cache = (key, Settings.from_env())
    return _cache[1]


# Knobs that modules freeze into module-level constants when they are first imported (a later change of the variable has no effect
# in that process); every other knob is read live through ``current()`` at the point of use.
IMPORT_TIME = frozenset((
    "debug_skip_trunk", "adapt_graph", "adapt_batched_trunk", "fused_dgrad", "fused_last_block", "x3_planes", "x3_fused_stats",
    "x3_fold_bn", "trunk_f16x2", "train_x3", "train_x3_min_rows", "fuse_next_c2_only", "fused_pair_mlp", "pair_f16x2", "pair_rk_rows", "gemm_rk_rows", "pair_mlp_gb", "wgrad_batch",
    "train_graph", "small_groups", "wf_xcd",
))
assert IMPORT_TIME <= set(k[0] for k in KNOBS), IMPORT_TIME - set(k[0] for k in KNOBS)


def describe():
    """The knob table as text (README / --help material): variable, field, when it is read, default, meaning."""
    return "\n".join("%-26s %-22s %-7s default %-8r %s" % (var, field, "import" if field in IMPORT_TIME else "live", default, doc)
                     for field, var, _, default, doc in KNOBS)
