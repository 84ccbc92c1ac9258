"""Meta-training driver (mirror of train.py:26-207 for --method gnnnet): Adam over all parameters, 100
episodes per epoch, checkpoints ``{'epoch','state'}`` under <save_dir>/checkpoints/<dataset>/<model>_<method>
[_aug]_<n>way_<k>shot/<epoch>.tar, optional first-order-MAML meta-fine-tuning (--fine_tune).
The episode source is the in-repo synthetic miniImageNet-shaped sampler (real data is out of scope)."""
import os

import numpy as np
import torch

from . import configs, optim, parallel, settings, synthetic
from .io_utils import get_assigned_file, model_dict, parse_args
from .methods import gnnnet_copy
from .methods.gnnnet import GnnNet


class SyntheticEpisodeLoader:
    """Stand-in for miniImageNet_few_shot.SetDataManager(...).get_data_loader(): ``n_episode`` episodes of
    [n_way, n_support+n_query, 3, size, size] (datasets/miniImageNet_few_shot.py:105-183)."""

    def __init__(self, n_way, n_support, n_query, size=84, n_episode=100, seed0=0, rank=0, world=1):
        self.a = (n_way, n_support, n_query, size)
        self.n_episode, self.seed0, self.epoch = n_episode, seed0, 0
        self.rank, self.world = rank, world

    def steps_per_rank(self):
        """Every rank MUST run the same number of steps: each step is one collective (AllReduceAdam.step), so a rank with one
        episode more would wait in an all-reduce its peers never enter.  floor(n_episode / W) steps per rank; the n_episode % W
        episodes left over at the end of the epoch's stream are not drawn (with W = 1 nothing is dropped)."""
        return self.n_episode // self.world

    def __len__(self):
        return self.steps_per_rank()

    def __iter__(self):
        base = self.seed0 + self.epoch * self.n_episode
        self.epoch += 1
        for s in range(self.steps_per_rank()):
            i = self.rank + s * self.world
            n_way, ns, nq, size = self.a
            yield synthetic.train_episode(base + i, n_way, ns, nq, size), None


class ResidentEpisodeLoader:
    """miniImageNet_few_shot.SetDataManager(size, n_query, n_way, n_support).get_data_loader(aug) over a dataset RESIDENT IN HBM
    (round-4 verdict "missing 3"): ``pool_u8`` [n_classes, n_per_class, Hs, Ws, 3] uint8 on the device.  Per episode, as the
    reference's loader does (datasets/miniImageNet_few_shot.py:53-74,105-107): the classes are ``randperm(n_classes)[:n_way]``
    (EpisodicBatchSampler) and every class contributes the first batch of a freshly shuffled DataLoader over its images, i.e.
    ``n_support + n_query`` distinct random images (SetDataset.__getitem__); each image then goes through the training-side
    transform (augment.sample_train_view_params: un-augmented box, or RandomResizedCrop + ImageJitter + horizontal flip with
    ``aug`` = --train_aug) in ONE launch of mft_augment_views that writes the normalised fp32 views -- no PIL, no host copy.
    The draws come from ``RandomState(f(seed, epoch, episode index))`` instead of torch's global stream, so that episode i is the
    same on every rank and rank r takes episodes r, r + W, ... (documented deviation, as parallel.episode_rng).
    Yields ``(x, None)`` with x [n_way, n_support + n_query, 3, size, size] on the device (channels-last strides: the kernel writes
    NHWC, the NCHW shape is a view of it)."""

    def __init__(self, pool_u8, n_way, n_support, n_query, size=84, n_episode=100, aug=False, seed=0, rank=0, world=1):
        if pool_u8.dtype != torch.uint8 or pool_u8.dim() != 5:
            raise ValueError("pool_u8 must be a uint8 tensor [n_classes, n_per_class, H, W, 3] (on the device: episode() launches "
                             "mft_augment_views on it and raises for a host tensor)")
        self.pool = pool_u8
        self.n_classes, self.n_per_class = pool_u8.shape[:2]
        if n_way > self.n_classes or n_support + n_query > self.n_per_class:
            raise ValueError("pool too small for %d-way episodes of %d images per class" % (n_way, n_support + n_query))
        self.a = (n_way, n_support, n_query, size)
        self.n_episode, self.aug, self.seed, self.epoch = n_episode, bool(aug), int(seed), 0
        self.rank, self.world = rank, world

    def steps_per_rank(self):
        return self.n_episode // self.world              # (same rule as SyntheticEpisodeLoader: one collective per step)

    def __len__(self):
        return self.steps_per_rank()

    def indices(self, epoch, i):
        """(classes [n_way], images [n_way, n_support + n_query], the episode's RandomState after those draws)."""
        n_way, ns, nq, _ = self.a
        rs = np.random.RandomState((self.seed * 1000003 + (epoch * self.n_episode + i) * 7919 + 17) % (2 ** 32 - 1))
        classes = rs.permutation(self.n_classes)[:n_way]
        images = np.stack([rs.permutation(self.n_per_class)[:ns + nq] for _ in range(n_way)])
        return classes, images, rs

    CHUNK = 64          # episodes whose index tables / view parameters travel to the device in ONE pinned, non-blocking copy

    def _tables(self, epoch, i):
        """Device tables of the chunk of episodes that contains episode i: (class index, image index) [CHUNK, n_way * per] and view
        parameters [CHUNK, 1, n_way * per, NPARAM].  A per-episode pageable host-to-device copy would drain the stream once per
        step (measured: +0.23 ms on a 3.9 ms meta-training step); a chunk costs one asynchronous copy per 64 episodes."""
        from . import augment
        c0 = (i // self.CHUNK) * self.CHUNK
        key = (epoch, c0)
        if getattr(self, "_chunk_key", None) != key:
            n_way, ns, nq, size = self.a
            per = ns + nq
            Hs, Ws = self.pool.shape[2], self.pool.shape[3]
            idx = np.empty((self.CHUNK, 2, n_way * per), dtype=np.int64)
            P = np.empty((self.CHUNK, 1, n_way * per, augment.NPARAM), dtype=np.float32)
            for j in range(self.CHUNK):
                classes, images, rs = self.indices(epoch, c0 + j)
                idx[j, 0] = np.repeat(classes, per)
                idx[j, 1] = images.reshape(-1)
                P[j] = augment.sample_train_view_params(rs, n_way * per, Hs, Ws, size, self.aug)
            dev = self.pool.device
            if dev.type == "cuda":
                self._chunk_idx = torch.from_numpy(idx).pin_memory().to(dev, non_blocking=True)
                self._chunk_P = torch.from_numpy(P).pin_memory().to(dev, non_blocking=True)
            else:
                self._chunk_idx, self._chunk_P = torch.from_numpy(idx), torch.from_numpy(P)
            self._chunk_key = key
        j = i - c0
        return self._chunk_idx[j, 0], self._chunk_idx[j, 1], self._chunk_P[j]

    def episode(self, epoch, i):
        from . import augment
        n_way, ns, nq, size = self.a
        per = ns + nq
        ci, ii, P = self._tables(epoch, i)
        src = self.pool[ci, ii].contiguous()                                 # [n_way * per, Hs, Ws, 3] uint8 gather
        v = augment.augment_views(src, P, size)                              # [1, n_way * per, size, size, 3] fp32, normalised
        return v.view(n_way, per, size, size, 3).permute(0, 1, 4, 2, 3)

    def __iter__(self):
        # (sampling episode s + 1 on a side stream while step s trains was measured SLOWER -- 4.35 vs 4.00 ms per meta-training step:
        #  the step is ~425 small kernels back to back, and a second queue's kernels and event hand-offs cost it more than the
        #  0.13 ms of gather + transform they would hide; the episode is produced on the consumer's stream.)
        epoch = self.epoch
        self.epoch += 1
        for s in range(self.steps_per_rank()):
            yield self.episode(epoch, self.rank + s * self.world), None


class SyntheticBatchLoader:
    """Stand-in for SimpleDataManager(image_size, batch_size=16).get_data_loader(aug) (train.py:105-108;
    datasets/miniImageNet_few_shot.py:146-163): ``n_batch`` supervised mini-batches (x [B,3,size,size], y [B]) over
    ``num_classes`` class-structured synthetic classes."""

    def __init__(self, num_classes, size=84, batch_size=16, n_batch=100, seed0=0):
        self.num_classes, self.size, self.bs, self.n_batch, self.seed0, self.epoch = num_classes, size, batch_size, n_batch, seed0, 0
        rs = np.random.RandomState(seed0 + 12345)
        low = torch.from_numpy(rs.standard_normal((num_classes, 3, 7, 7)).astype(np.float32))
        self.templates = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)

    def __len__(self):
        return self.n_batch

    def __iter__(self):
        rs = np.random.RandomState(self.seed0 + self.epoch)
        self.epoch += 1
        for _ in range(self.n_batch):
            y = rs.randint(0, self.num_classes, size=self.bs)
            x = self.templates[torch.from_numpy(y)] + torch.from_numpy(rs.standard_normal((self.bs, 3, self.size, self.size)).astype(np.float32))
            yield x, torch.from_numpy(y)


class AllReduceAdam:
    """Episode-parallel meta-training (new functionality: the reference is one process): every rank has run
    ONE episode's backward from the common parameters; ``step`` sums all ranks' gradients in a single flat fp32 bucket
    all-reduce (RCCL over xGMI under torchrun; 21.2 MB for GnnNet), divides by the world size and applies the same fused
    Adam step on every rank.  With one rank it is the plain optimiser.  MetaTemplate's loops call only zero_grad() / step()."""

    def __init__(self, params_iter, **kw):
        ps = list(params_iter)
        self.opt = optim.Adam(ps, **kw)          # torch.optim.Adam semantics, fused HIP update (train.py:28)
        self.bucket = parallel.FlatGradBucket(ps) if (parallel.world()[1] > 1 or parallel.collectives_forced()) else None
        self.param_groups, self.state = self.opt.param_groups, self.opt.state

    def zero_grad(self, *a, **k):
        return self.opt.zero_grad(*a, **k)

    def step(self, closure=None):
        if self.bucket is not None:
            W = self.bucket.allreduce_sum()          # .grad = SUM over the ranks; the fused Adam launch reads it times 1 / W
            self.opt.grad_scale = 1.0 / W
        return self.opt.step(closure)


def train(base_loader, model, optimization, start_epoch, stop_epoch, params, variant50=False):
    """train.py:26-61 (``variant50``: train_50.py:34-71 -- the 50-shot loops and a checkpoint every 10 epochs)."""
    if optimization != 'Adam':
        raise ValueError('Unknown optimization, please define by yourself')
    optimizer = AllReduceAdam(model.parameters())
    rank, W = parallel.world()
    fifty = variant50 and params.n_shot == 50
    for epoch in range(start_epoch, stop_epoch):
        model.train()
        if params.method == 'baseline':
            model.train_loop(epoch, base_loader, optimizer)              # train.py:41-42: every other method -> train_loop
        elif not params.fine_tune and getattr(params, "episodes_per_rank", 1) > 1:
            if fifty:
                raise NotImplementedError("--episodes_per_rank with the 50-shot loops")
            model.train_loop_lockstep(epoch, base_loader, optimizer, params.episodes_per_rank)
        elif not params.fine_tune:
            (model.train_loop50 if fifty else model.train_loop2)(epoch, base_loader, optimizer)
        else:
            (model.train_loop_finetune50 if fifty else model.train_loop_finetune)(epoch, base_loader, optimizer)
            if epoch == (stop_epoch - 1):
                model.MAML_update()
        if not os.path.isdir(params.checkpoint_dir):
            os.makedirs(params.checkpoint_dir, exist_ok=True)
        save_freq = 10 if variant50 else params.save_freq                # train_50.py:53,66
        if (epoch % save_freq == 0) or (epoch == stop_epoch - 1):
            # BatchNorm running statistics diverge per rank (never read on the hot path): the checkpoint takes rank 0's
            parallel.broadcast_buffers(model, src=0)
            if rank == 0:
                outfile = os.path.join(params.checkpoint_dir, '{:d}.tar'.format(epoch))
                torch.save({'epoch': epoch, 'state': model.state_dict()}, outfile)
    return model


def main(argv=None, n_episode=100, size=84, variant50=False, pool_images_per_class=None):
    """``--dataset miniImageNet`` (the only one the reference's few-shot branch accepts, train.py:116-125): episodes are sampled
    on the device from a miniImageNet-SHAPED resident pool (synthetic.class_pool_u8: 64 classes x 600 images x 84x84 uint8) by
    ResidentEpisodeLoader, ``--train_aug`` selecting the augmenting transform; any other --dataset value keeps the host-side
    SyntheticEpisodeLoader (fp32 episodes drawn per seed)."""
    params = parse_args('train', argv)
    from .finetune import _init_distributed
    parallel.limit_host_threads()
    _init_distributed()                                                  # under torchrun: one process per GPU, RCCL
    rank, W = parallel.world()
    if not params.start_epoch > 0:
        np.random.seed(10)
    if params.method not in ('gnnnet', 'baseline'):
        raise NotImplementedError("--method %s: 'gnnnet' and 'baseline' are on the HIP path" % params.method)
    params.checkpoint_dir = '%s/checkpoints/%s/%s_%s' % (configs.save_dir, params.dataset, params.model, params.method)
    if params.train_aug:
        params.checkpoint_dir += '_aug'
    if params.method == 'baseline':                                      # train.py:101-108,176-180
        from .methods.baselinetrain import BaselineTrain
        base_loader = SyntheticBatchLoader(params.num_classes, size, 16, n_episode)
        model = BaselineTrain(model_dict[params.model], params.num_classes).cuda()
    else:
        n_query = max(1, int(16 * params.test_n_way / params.train_n_way))
        # every rank draws its own episodes (rank r takes episode r, r+W, ... of the epoch's stream); same model init on all
        if params.dataset == "miniImageNet" and settings.current().train_source == "pool":
            pool = synthetic.class_pool_u8("miniImageNet", torch.device("cuda", torch.cuda.current_device()), seed=0,
                                           n_per_class=pool_images_per_class)
            base_loader = ResidentEpisodeLoader(pool, params.train_n_way, params.n_shot, n_query, size, n_episode,
                                                aug=params.train_aug, rank=rank, world=W)
        else:
            base_loader = SyntheticEpisodeLoader(params.train_n_way, params.n_shot, n_query, size, n_episode, rank=rank, world=W)
        cls = gnnnet_copy.GnnNet if (variant50 and params.n_shot == 50) else GnnNet          # train_50.py:154-157
        torch.manual_seed(0) if W > 1 else None
        model = cls(model_dict[params.model], n_way=params.train_n_way, n_support=params.n_shot).cuda()
        params.checkpoint_dir += '_%dway_%dshot' % (params.train_n_way, params.n_shot)
    os.makedirs(params.checkpoint_dir, exist_ok=True)
    if params.start_epoch > 0:
        tmp = torch.load(get_assigned_file(params.checkpoint_dir, params.start_epoch - 1))
        state = {k: v for k, v in tmp['state'].items() if "feature2." not in k and "feature3." not in k}
        model.load_state_dict(state)
    if W > 1:
        parallel.broadcast_parameters(model, src=0)                      # identical initial parameters on every rank
    return train(base_loader, model, 'Adam', params.start_epoch, params.stop_epoch, params, variant50=variant50)


if __name__ == '__main__':
    main()
