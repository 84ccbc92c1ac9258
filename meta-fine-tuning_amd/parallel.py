"""Multi-GPU layer: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm, "gloo"
in the CPU tests).  The reference is single-process; this is new functionality with two modes:

* test-time fine-tuning shards embarrassingly: episode i -> rank i mod W, no data-path collective; one
  all-gather of per-episode accuracies at the end.  Each episode draws its permutations from its own
  ``numpy.random.RandomState(f(seed, i))`` so results do not depend on the rank count (deviation from the
  reference's single sequential stream).
* meta-training has ONE exchange step: every rank runs one episode from the common parameters, the gradients of
  all 5,307,706 parameters are summed in a single flat fp32 bucket all-reduce (21.2 MB; ring all-reduce moves
  2*(W-1)/W of it per GPU, ~0.25 ms on one 153 GB/s xGMI link) and divided by W, then every rank applies the same
  outer Adam step.  Parity oracle: accumulate W episodes' gradients sequentially, divide by W, one step.
"""
import numpy as np
import torch
import torch.distributed as dist


def host_threads(cap=32):
    """CPU threads this process may really use: affinity mask and cgroup CPU quota (a GPU box typically grants a container 16 of
    its 256 cores), capped at ``cap``."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


_THREADS_LIMITED = False


def limit_host_threads():
    """torch sizes its intra-op pool by os.cpu_count() (256 on an MI355X host); under a 16-core cgroup quota the first parallel CPU
    op of the process then stalls for seconds (measured: 1-3 s of a 600-episode evaluation's fixed cost).  The drivers' host work is
    a few small tensor ops: cap the pool at the granted cores, divided over the ranks of this node.  OMP_NUM_THREADS (the user's
    choice, and what torchrun sets) wins; idempotent."""
    global _THREADS_LIMITED
    import os
    if _THREADS_LIMITED or "OMP_NUM_THREADS" in os.environ:
        return
    _THREADS_LIMITED = True
    W = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    n = max(1, host_threads(16) // max(1, W))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collectives_forced():
    """MFT_FORCE_COLLECTIVES=1 with an initialised process group: run the collectives even with ONE rank instead of taking the
    single-process shortcuts -- so that a one-GPU box executes the real RCCL calls (communicator set-up, all-reduce of the flat
    gradient bucket, all-gather, broadcast) on the tensors the multi-GPU drivers use (tests/test_drivers_gpu.py)."""
    from . import settings
    return settings.current().force_collectives and dist.is_available() and dist.is_initialized()


def shard_indices(n_items, rank, world_size):
    """Episode i belongs to rank i mod W."""
    return list(range(rank, n_items, world_size))


def episode_rng(seed, episode_index):
    """Rank-count-invariant permutation stream of one episode."""
    return np.random.RandomState((int(seed) * 1000003 + int(episode_index) * 7919) % (2 ** 32 - 1))


def episode_torch_seed(seed, episode_index):
    """Seed of the torch generator that draws episode i's Classifier initialisation (finetune.py:65) in sharded runs."""
    return (int(seed) * 1000003 + int(episode_index) * 7919 + 104729) % (2 ** 31 - 1)


def gather_episode_values(local_values, n_items, device="cpu"):
    """All ranks' per-episode values (sharded with ``shard_indices``) -> full array in episode order on every rank."""
    rank, W = world()
    if W == 1 and not collectives_forced():
        return np.asarray(local_values, dtype=np.float64)
    per = (n_items + W - 1) // W
    buf = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
    buf[:len(local_values)] = torch.as_tensor(np.asarray(local_values, dtype=np.float64), device=device)
    out = [torch.empty_like(buf) for _ in range(W)]
    dist.all_gather(out, buf)
    full = np.empty(n_items, dtype=np.float64)
    for r in range(W):
        idx = shard_indices(n_items, r, W)
        full[idx] = out[r][:len(idx)].cpu().numpy()
    return full


class FlatGradBucket:
    """One flat fp32 buffer holding every parameter's gradient: a single all-reduce per outer step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        # (+ one float per parameter behind the gradients: 1 where this rank holds a gradient -- summed with them, so that ranks
        # that disagree about WHICH parameters have one are caught instead of silently applying different updates: ADVICE r05)
        self.flat_all = torch.zeros(self.numel + len(self.params), dtype=torch.float32, device=p0.device)
        self.flat = self.flat_all[:self.numel]
        self.has = self.flat_all[self.numel:]
        self.calls = 0
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def allreduce_sum(self):
        """``allreduce_mean`` without the division: ``.grad`` holds the SUM over the ranks afterwards and the caller's optimiser
        applies 1 / W as it reads the gradients (optim.Adam.grad_scale) -- one launch less per step.  Returns W."""
        return self.allreduce_mean(divide=False)

    def allreduce_mean(self, divide=True):
        """Pack grads -> all-reduce(SUM) -> divide by W -> unpack into .grad (in place).  With one rank nothing is exchanged
        and the gradients stay where autograd put them.  A parameter whose ``.grad`` is None (frozen this step; every rank
        freezes the same ones) contributes zeros to the sum and KEEPS ``grad = None``, so the optimiser skips it as
        torch.optim.Adam does -- no per-step ``zeros_like`` allocation, and the addresses in optim.Adam's pointer table stay put
        (ADVICE r04)."""
        _, W = world()
        if W == 1 and not collectives_forced():
            return W
        live = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        # every rank checks on the same schedule (the first two calls, then every 64th): the collective's size must agree across ranks
        check = self.calls < 2 or self.calls % 64 == 0
        self.calls += 1
        if check:
            self.has.copy_(torch.tensor([p.grad is not None for p in self.params], dtype=torch.float32))
        if len(live) != len(self.params):
            for v, p in zip(self.views, self.params):
                if p.grad is None:
                    v.zero_()
        views, grads = [v for v, _ in live], [g for _, g in live]
        if views:
            torch._foreach_copy_(views, grads)                  # fused multi-tensor copies (104 tensors -> a few launches)
        buf = self.flat_all if check else self.flat               # (check steps: the has-gradient floats ride behind the gradients)
        if buf.is_cuda and dist.get_backend() == "gloo":          # CPU-test / one-device hook: stage through the host
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)            # "nccl" == RCCL over xGMI
        if check:
            got = self.has.cpu()
            bad = [i for i, v in enumerate(got.tolist()) if v not in (0.0, float(W))]
            if bad:
                raise RuntimeError("FlatGradBucket: the ranks disagree about which parameters have a gradient (parameter indices %s: "
                                   "%s of %d ranks hold one) -- a rank-dependent frozen / unused parameter would make the ranks apply "
                                   "different updates" % (bad[:8], [int(got[i]) for i in bad[:8]], W))
        if divide:
            self.flat.div_(W)
        if views:
            torch._foreach_copy_(grads, views)
        return W


def broadcast_buffers(module, src=0):
    """BatchNorm running statistics diverge per rank (each saw its own episodes) and are never read on the hot path;
    checkpoints take rank ``src``'s copy."""
    _, W = world()
    if W == 1 and not collectives_forced():
        return
    for b in module.buffers():
        if b.is_cuda and dist.get_backend() == "gloo":
            host = b.cpu()
            dist.broadcast(host, src=src)
            b.copy_(host)
        else:
            dist.broadcast(b, src=src)


def broadcast_parameters(module, src=0):
    """Rank ``src``'s parameters to every rank (once, before the first step: all ranks then stay identical because they apply
    the same averaged gradient)."""
    _, W = world()
    if W == 1 and not collectives_forced():
        return
    with torch.no_grad():
        for p in module.parameters():
            if p.is_cuda and dist.get_backend() == "gloo":
                host = p.detach().cpu()
                dist.broadcast(host, src=src)
                p.copy_(host)
            else:
                dist.broadcast(p.data, src=src)
