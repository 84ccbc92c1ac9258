"""Device-side test-time views: host parameter sampler + launcher for ``mft_augment_views``.

View contract of the reference (datasets/EuroSAT_few_shot.py:240-276 ``SubDataset2``): every episode image yields
``2 + num_aug`` tensors -- two identical un-augmented views (Scale(1.15*size) + CenterCrop(size)) followed by ``num_aug``
augmented ones (RandomSizedCrop(size, scale=(0.5, 0.9)), ImageJitter(Brightness .1, Contrast .1, Color .05),
RandomHorizontalFlip, RandomVerticalFlip), all ToTensor + Normalize(ImageNet).  The transforms themselves live in
torchvision (pinned 0.8.2, requirements.txt:20; not vendored in the reference); their published parameter draws are
restated here with numpy's RandomState so that an episode's views are a pure function of (seed, episode index):

* RandomSizedCrop == RandomResizedCrop.get_params: up to 10 tries of area ~ U(scale)*H*W, log-uniform aspect in [3/4, 4/3],
  w = round(sqrt(area*aspect)), h = round(sqrt(area/aspect)), accepted when it fits, top-left uniform; fallback = the
  largest centred crop within the aspect bounds.
* ImageJitter (data/additional_transforms.py:21-31): r_k = alpha_k*(2u-1)+1 for Brightness, Contrast, Color in that order.
* flips with probability 1/2 each.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
JITTER = (0.1, 0.1, 0.05)          # Brightness, Contrast, Color (EuroSAT_few_shot.py:131)
NPARAM = 10


def noaug_box(Hs, Ws, size):
    """Scale([int(1.15*size)]*2) then CenterCrop(size), expressed as the source-pixel box that lands on the size x size output."""
    S2 = int(size * 1.15)
    top = int(round((S2 - size) / 2.0))
    return top * Hs / S2, top * Ws / S2, size * Hs / S2, size * Ws / S2


def random_resized_crop_boxes(rs, n, Hs, Ws, scale=(0.5, 0.9), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """n independent RandomResizedCrop.get_params draws, vectorised: all 10 candidate boxes of every draw are generated at
    once and the first that fits is taken (same acceptance rule as the sequential loop) -> float32 [n, 4] (y0, x0, h, w)."""
    area = float(Hs * Ws)
    target = rs.uniform(scale[0], scale[1], size=(n, 10)) * area
    aspect = np.exp(rs.uniform(math.log(ratio[0]), math.log(ratio[1]), size=(n, 10)))
    w = np.rint(np.sqrt(target * aspect)).astype(np.int64)
    h = np.rint(np.sqrt(target / aspect)).astype(np.int64)
    ok = (w > 0) & (w <= Ws) & (h > 0) & (h <= Hs)
    first = np.argmax(ok, axis=1)
    any_ok = ok.any(axis=1)
    idx = np.arange(n)
    w, h = w[idx, first], h[idx, first]
    # fallback (no candidate fits): the largest centred crop within the aspect bounds
    in_ratio = Ws / Hs
    if in_ratio < ratio[0]:
        fw, fh = Ws, int(round(Ws / ratio[0]))
    elif in_ratio > ratio[1]:
        fh, fw = Hs, int(round(Hs * ratio[1]))
    else:
        fw, fh = Ws, Hs
    w = np.where(any_ok, w, fw)
    h = np.where(any_ok, h, fh)
    u = rs.uniform(0.0, 1.0, size=(n, 2))
    i = np.where(any_ok, np.floor(u[:, 0] * (Hs - h + 1)), (Hs - h) // 2)
    j = np.where(any_ok, np.floor(u[:, 1] * (Ws - w + 1)), (Ws - w) // 2)
    return np.stack([i, j, h, w], axis=1).astype(np.float32)


def sample_view_params(rs, n_img, Hs, Ws, size, num_aug):
    """-> float32 [2 + num_aug, n_img, 10] rows (y0, x0, h, w, r_brightness, r_contrast, r_color, flip_h, flip_v, enhance)."""
    P = np.zeros((2 + num_aug, n_img, NPARAM), dtype=np.float32)
    P[:2, :, 0:4] = noaug_box(Hs, Ws, size)
    P[:2, :, 4:7] = 1.0
    if num_aug > 0:
        n = num_aug * n_img
        P[2:, :, 0:4] = random_resized_crop_boxes(rs, n, Hs, Ws).reshape(num_aug, n_img, 4)
        u = rs.uniform(0.0, 1.0, size=(n, 3))
        P[2:, :, 4:7] = (np.asarray(JITTER)[None] * (2.0 * u - 1.0) + 1.0).reshape(num_aug, n_img, 3)
        P[2:, :, 7:9] = (rs.uniform(0.0, 1.0, size=(n, 2)) < 0.5).reshape(num_aug, n_img, 2)
        P[2:, :, 9] = 1.0
    return P


TRAIN_JITTER = (0.4, 0.4, 0.4)     # Brightness, Contrast, Color (datasets/miniImageNet_few_shot.py:108)


def sample_train_view_params(rs, n_img, Hs, Ws, size, aug):
    """Training-side transform of the reference's SetDataManager (datasets/miniImageNet_few_shot.py:109-141,178), ONE view per
    image -> float32 [1, n_img, 10].  ``aug`` False: Resize([int(1.15 * size)] * 2) + CenterCrop(size) (the un-augmented box);
    True (--train_aug): RandomResizedCrop(size) with torchvision's default scale (0.08, 1) and ratio (3/4, 4/3),
    ImageJitter(Brightness .4, Contrast .4, Color .4), RandomHorizontalFlip (no vertical flip on this side)."""
    P = np.zeros((1, n_img, NPARAM), dtype=np.float32)
    P[0, :, 4:7] = 1.0
    if not aug:
        P[0, :, 0:4] = noaug_box(Hs, Ws, size)
        return P
    P[0, :, 0:4] = random_resized_crop_boxes(rs, n_img, Hs, Ws, scale=(0.08, 1.0))
    u = rs.uniform(0.0, 1.0, size=(n_img, 3))
    P[0, :, 4:7] = np.asarray(TRAIN_JITTER)[None] * (2.0 * u - 1.0) + 1.0
    P[0, :, 7] = rs.uniform(0.0, 1.0, size=n_img) < 0.5
    P[0, :, 9] = 1.0
    return P


def sample_view_params_torch(n_img, Hs, Ws, size, num_aug):
    """``sample_view_params`` on the REFERENCE's random stream: every draw comes from torch's global generator in the order
    the reference's loader consumes it -- image by image (SubDataset2.__getitem__, datasets/EuroSAT_few_shot.py:156-170), per
    augmented view RandomSizedCrop.get_params (torchvision 0.8.2: up to 10 x (uniform_ area, uniform_ log-aspect), then two
    randint), ImageJitter's torch.rand(3) (data/additional_transforms.py:24), the two flips' torch.rand(1) -- so that after
    torch.manual_seed(s) the views are the ones the reference's transforms would produce for the same images.
    Sequential and therefore slow (~8 small torch calls per view); ``sample_view_params`` is the vectorised numpy-stream form."""
    P = np.zeros((2 + num_aug, n_img, NPARAM), dtype=np.float32)
    P[:2, :, 0:4] = noaug_box(Hs, Ws, size)
    P[:2, :, 4:7] = 1.0
    area = Hs * Ws
    lo, hi = math.log(3.0 / 4.0), math.log(4.0 / 3.0)
    for n in range(n_img):
        for a in range(num_aug):
            box = None
            for _ in range(10):
                target_area = area * torch.empty(1).uniform_(0.5, 0.9).item()
                log_ratio = torch.log(torch.tensor((3.0 / 4.0, 4.0 / 3.0)))
                aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
                w = int(round(math.sqrt(target_area * aspect)))
                h = int(round(math.sqrt(target_area / aspect)))
                if 0 < w <= Ws and 0 < h <= Hs:
                    i = torch.randint(0, Hs - h + 1, size=(1,)).item()
                    j = torch.randint(0, Ws - w + 1, size=(1,)).item()
                    box = (i, j, h, w)
                    break
            if box is None:
                in_ratio = float(Ws) / float(Hs)
                if in_ratio < 3.0 / 4.0:
                    w, h = Ws, int(round(Ws / (3.0 / 4.0)))
                elif in_ratio > 4.0 / 3.0:
                    h, w = Hs, int(round(Hs * (4.0 / 3.0)))
                else:
                    w, h = Ws, Hs
                box = ((Hs - h) // 2, (Ws - w) // 2, h, w)
            r = torch.rand(3)
            P[2 + a, n, 0:4] = box
            for k in range(3):
                P[2 + a, n, 4 + k] = float(JITTER[k] * (r[k] * 2.0 - 1.0) + 1)
            P[2 + a, n, 7] = float(bool(torch.rand(1) < 0.5))
            P[2 + a, n, 8] = float(bool(torch.rand(1) < 0.5))
            P[2 + a, n, 9] = 1.0
    return P


def augment_views(src_u8, params, size, out=None, view_stride=None, img_stride=None):
    """src_u8 [n_img, Hs, Ws, 3] uint8 device tensor, params [n_views, n_img, 10] (numpy or tensor) ->
    fp32 NHWC views [n_views, n_img, size, size, 3] (or written into ``out`` with the given strides, in floats)."""
    if not src_u8.is_cuda or src_u8.dtype != torch.uint8 or not src_u8.is_contiguous():
        raise RuntimeError("augment_views needs a contiguous uint8 CUDA (HIP) tensor [n_img, Hs, Ws, 3]")
    n_img, Hs, Ws, _ = src_u8.shape
    if torch.is_tensor(params) and params.is_cuda and params.dtype == torch.float32:
        p = params.contiguous()                  # already on the device (train.ResidentEpisodeLoader uploads a chunk of episodes at once)
    else:
        p = torch.as_tensor(params, dtype=torch.float32).to(src_u8.device).contiguous()
    n_views = p.shape[0]
    assert p.shape == (n_views, n_img, NPARAM)
    if out is None:
        out = torch.empty((n_views, n_img, size, size, 3), device=src_u8.device, dtype=torch.float32)
        view_stride, img_stride = n_img * size * size * 3, size * size * 3
    mean = (ctypes.c_float * 3)(*IMAGENET_MEAN)
    std = (ctypes.c_float * 3)(*IMAGENET_STD)
    rc = _lib.lib().mft_augment_views(ops._p(src_u8), n_img, Hs, Ws, ops._p(p), n_views, ops._p(out), view_stride, img_stride,
                                      size, ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(std, ctypes.c_void_p), ops._stream())
    _lib.check(rc, "mft_augment_views")
    return out


def episode_views(src_u8, n_way, n_per_class, size, num_aug, rs):
    """The reference's episode contract from raw images: src_u8 [n_way*n_per_class, Hs, Ws, 3] (class-major) ->
    list of 2 + num_aug NHWC tensors [n_way, n_per_class, size, size, 3] (views 0 and 1 identical)."""
    n_img, Hs, Ws, _ = src_u8.shape
    assert n_img == n_way * n_per_class
    P = sample_view_params(rs, n_img, Hs, Ws, size, num_aug)
    v = augment_views(src_u8, P, size)
    return [v[i].view(n_way, n_per_class, size, size, 3) for i in range(2 + num_aug)], P


class EpisodeSampler:
    """Test-time episode source over a dataset resident in HBM (second half).

    Reference semantics (datasets/EuroSAT_few_shot.py:75-124,187-205,329-351): an episode's classes are
    ``torch.randperm(n_classes)[:n_way]`` (EpisodicBatchSampler2.generate_perm) and, per class, the first batch of a freshly
    shuffled DataLoader over that class's images, i.e. ``n_support + n_query`` distinct random images; every image then
    yields the 2 + num_aug views.  Here the uint8 images live on the device as ``data[n_classes, n_per_class, H, W, 3]``;
    the draws come from ``numpy.random.RandomState(f(seed, episode index))`` (not torch's global stream), so episode i is
    the same on every rank and for every batch size -- a documented deviation, like parallel.episode_rng."""

    def __init__(self, data_u8, n_way=5, n_per_episode=20, seed=7):
        if data_u8.dtype != torch.uint8 or data_u8.dim() != 5:
            raise ValueError("data_u8 must be uint8 [n_classes, n_per_class, H, W, 3]")
        self.data = data_u8
        self.n_classes, self.n_per_class = data_u8.shape[:2]
        if n_way > self.n_classes or n_per_episode > self.n_per_class:
            raise ValueError("dataset too small for %d-way episodes of %d images per class" % (n_way, n_per_episode))
        self.n_way, self.n_per_episode, self.seed = n_way, n_per_episode, seed

    def indices(self, episode):
        rs = np.random.RandomState((int(self.seed) * 1000003 + int(episode) * 7919) % (2 ** 32 - 1))
        classes = rs.permutation(self.n_classes)[:self.n_way]
        images = np.stack([rs.permutation(self.n_per_class)[:self.n_per_episode] for _ in range(self.n_way)])
        return classes, images, rs

    def episode(self, episode, size, num_aug):
        """-> (src_u8 [n_way, n_per_episode, H, W, 3] on the data's device, view params [2+num_aug, n_way*n_per_episode, 10],
        classes): exactly what FinetuneEngine.run_batch(..., sources=True) ingests."""
        classes, images, rs = self.indices(episode)
        ci = torch.from_numpy(np.repeat(classes, self.n_per_episode)).to(self.data.device)
        ii = torch.from_numpy(images.reshape(-1)).to(self.data.device)
        src = self.data[ci, ii].view(self.n_way, self.n_per_episode, *self.data.shape[2:]).contiguous()
        P = sample_view_params(rs, self.n_way * self.n_per_episode, self.data.shape[2], self.data.shape[3], size, num_aug)
        return src, P, classes
