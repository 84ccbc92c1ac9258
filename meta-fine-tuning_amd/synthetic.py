"""Deterministic synthetic weights and episodes.

Nothing here comes from the reference's RNG: weights and images are drawn from
``numpy.random.RandomState`` (frozen MT19937 stream) so the same seed gives the
same tensors in this container, on the GPU box, and inside the oracle scripts.

* weights follow the reference's init *distributions*
  (conv: N(0, sqrt(2/(k*k*C_out)))  -- backbone.py:9-13;
   BatchNorm: gamma=1, beta=0        -- backbone.py:14-16;
   nn.Linear / GNN 1x1 nn.Conv2d: torch default U(-1/sqrt(fan_in), 1/sqrt(fan_in)))
  with an optional perturbation of the BatchNorm affine terms so that parity
  tests can tell gamma from beta.
* key names / shapes follow SURVEY.md Appendix A (``GnnNet(ResNet10, n_way, k)``).
* episodes follow the data contract of SURVEY.md §3.4: fp32 NCHW
  ``[n_way, n_support + n_query, 3, H, H]``; test-time episodes are a list of
  ``2 + gen_examples`` views whose first two entries are identical
  (finetune.py:606 asserts that).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

RESNET10_STAGES = [(64, 64, False), (64, 128, True), (128, 256, True), (256, 512, True)]


def _conv_w(rs, cout, cin, k):
    std = math.sqrt(2.0 / float(k * k * cout))
    return torch.from_numpy((rs.standard_normal((cout, cin, k, k)) * std).astype(np.float32))


def _uniform(rs, shape, bound):
    return torch.from_numpy(rs.uniform(-bound, bound, size=shape).astype(np.float32))


def _bn(rs, sd, prefix, c, perturb, buffers=True):
    if perturb:
        sd[prefix + ".weight"] = torch.from_numpy(rs.uniform(0.5, 1.5, size=(c,)).astype(np.float32))
        sd[prefix + ".bias"] = torch.from_numpy((rs.standard_normal((c,)) * 0.1).astype(np.float32))
    else:
        sd[prefix + ".weight"] = torch.ones(c)
        sd[prefix + ".bias"] = torch.zeros(c)
    if buffers:
        sd[prefix + ".running_mean"] = torch.zeros(c)
        sd[prefix + ".running_var"] = torch.ones(c)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def resnet10_state_dict(seed=0, perturb_bn=True, prefix=""):
    """State dict of backbone.ResNet10(flatten=True) in reference key order."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    sd[prefix + "trunk.0.weight"] = _conv_w(rs, 64, 3, 7)
    _bn(rs, sd, prefix + "trunk.1", 64, perturb_bn)
    for i, (cin, cout, half) in enumerate(RESNET10_STAGES):
        p = prefix + "trunk.%d" % (4 + i)
        sd[p + ".C1.weight"] = _conv_w(rs, cout, cin, 3)
        _bn(rs, sd, p + ".BN1", cout, perturb_bn)
        sd[p + ".C2.weight"] = _conv_w(rs, cout, cout, 3)
        _bn(rs, sd, p + ".BN2", cout, perturb_bn)
        if cin != cout:
            sd[p + ".shortcut.weight"] = _conv_w(rs, cout, cin, 1)
            _bn(rs, sd, p + ".BNshortcut", cout, perturb_bn)
    return sd


def gnn_head_state_dict(seed=1, n_way=5, perturb_bn=True):
    """fc.* and gnn.* entries of GnnNet (methods/gnnnet.py:30-31, methods/gnn.py:134-152)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    sd["fc.0.weight"] = _uniform(rs, (128, 512), 1.0 / math.sqrt(512))
    sd["fc.0.bias"] = _uniform(rs, (128,), 1.0 / math.sqrt(512))
    _bn(rs, sd, "fc.1", 128, perturb_bn, buffers=False)
    nf = 96
    f0 = 128 + n_way

    def wcompute(name, fin):
        dims = [(fin, 2 * nf), (2 * nf, 2 * nf), (2 * nf, nf), (nf, nf)]
        for li, (a, b) in enumerate(dims, start=1):
            sd["%s.conv2d_%d.weight" % (name, li)] = _uniform(rs, (b, a, 1, 1), 1.0 / math.sqrt(a))
            sd["%s.conv2d_%d.bias" % (name, li)] = _uniform(rs, (b,), 1.0 / math.sqrt(a))
            _bn(rs, sd, "%s.bn_%d" % (name, li), b, perturb_bn, buffers=False)
        sd[name + ".conv2d_last.weight"] = _uniform(rs, (1, nf, 1, 1), 1.0 / math.sqrt(nf))
        sd[name + ".conv2d_last.bias"] = _uniform(rs, (1,), 1.0 / math.sqrt(nf))

    def gconv(name, fin, fout, bn):
        sd[name + ".fc.weight"] = _uniform(rs, (fout, 2 * fin), 1.0 / math.sqrt(2 * fin))
        sd[name + ".fc.bias"] = _uniform(rs, (fout,), 1.0 / math.sqrt(2 * fin))
        if bn:
            _bn(rs, sd, name + ".bn", fout, perturb_bn, buffers=False)

    for i in range(2):
        fin = f0 + (nf // 2) * i
        wcompute("gnn.layer_w%d" % i, fin)
        gconv("gnn.layer_l%d" % i, fin, nf // 2, True)
    fin = f0 + (nf // 2) * 2
    wcompute("gnn.w_comp_last", fin)
    gconv("gnn.layer_last", fin, n_way, False)
    return sd


def gnnnet_state_dict(seed=0, n_way=5, perturb_bn=True):
    """Full 140-entry state dict of GnnNet(ResNet10, n_way, k)."""
    sd = resnet10_state_dict(seed, perturb_bn, prefix="feature.")
    sd.update(gnn_head_state_dict(seed + 1, n_way, perturb_bn))
    return sd


def gnnnet_state_dict_with_running_stats(seed=0, n_way=5):
    """``gnnnet_state_dict`` whose BatchNorm running statistics are not the (0, 1) defaults, so that eval-mode paths
    (finetune(freeze_backbone=True), finetune.py:262-266) provably read them."""
    sd = gnnnet_state_dict(seed, n_way)
    rs = np.random.RandomState(seed + 1)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy((rs.standard_normal(tuple(sd[k].shape)) * 0.2).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, size=tuple(sd[k].shape)).astype(np.float32))
    return sd


# ----------------------------------------------------------------------------- episodes

def _templates(rs, n_way, size):
    """Per-class low-frequency templates: 7x7 gaussian grids upsampled bilinearly."""
    low = torch.from_numpy(rs.standard_normal((n_way, 3, 7, 7)).astype(np.float32))
    return torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)


def train_episode(seed, n_way=5, n_support=5, n_query=16, size=84, structured=True, noise=1.0):
    """One meta-train episode ``[n_way, n_support+n_query, 3, size, size]`` fp32."""
    rs = np.random.RandomState(seed)
    n = n_support + n_query
    if structured:
        t = _templates(rs, n_way, size)
        x = t[:, None] + noise * torch.from_numpy(rs.standard_normal((n_way, n, 3, size, size)).astype(np.float32))
    else:
        x = torch.from_numpy(rs.uniform(0.0, 1.0, size=(n_way, n, 3, size, size)).astype(np.float32))
    return x.contiguous()


def test_episode(seed, n_way=5, n_support=5, n_query=15, size=84, gen_examples=17,
                 structured=True, noise=1.0, aug_noise=0.1):
    """Test-time episode: list of ``2 + gen_examples`` views.

    Views 0 and 1 are bit-identical (finetune.py:606); views 2.. are the base
    images plus small noise, horizontally flipped for odd view index -- a
    stand-in for the reference's PIL augmentation pipeline
    (datasets/EuroSAT_few_shot.py:145-170), which is out of scope.
    """
    rs = np.random.RandomState(seed)
    base = train_episode(rs.randint(0, 2 ** 31 - 1), n_way, n_support, n_query, size, structured, noise)
    views = [base, base.clone()]
    for g in range(gen_examples):
        v = base + aug_noise * torch.from_numpy(rs.standard_normal(tuple(base.shape)).astype(np.float32))
        if g % 2 == 1:
            v = torch.flip(v, dims=[-1])
        views.append(v.contiguous())
    return views


def episode_labels(n_way, n):
    """Implicit labels ``np.repeat(range(n_way), n)`` (gnnnet.py:119,220; finetune.py:217)."""
    return np.repeat(np.arange(n_way), n)


def test_episode_device(seed, device, n_way=5, n_support=5, n_query=15, size=84, gen_examples=17, noise=1.0,
                        aug_noise=0.1):
    """Same contract as ``test_episode`` but drawn with torch's device generator straight into HBM (used by bench.py
    to keep start-up short: 19 views x 100 images x 84 KB = 160 MB per episode).  Deterministic per (seed, device type);
    NOT bit-identical to the numpy version."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    low = torch.randn((n_way, 3, 7, 7), generator=g, device=device)
    t = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)
    n = n_support + n_query
    base = (t[:, None] + noise * torch.randn((n_way, n, 3, size, size), generator=g, device=device)).contiguous()
    views = [base, base.clone()]
    for k in range(gen_examples):
        v = base + aug_noise * torch.randn(base.shape, generator=g, device=device)
        if k % 2 == 1:
            v = torch.flip(v, dims=[-1])
        views.append(v.contiguous())
    return views


# dataset shapes: classes x images per class x source side.  miniImageNet: 64 base classes
# (datasets/miniImageNet_few_shot.py:53) x 600 images, served downsampled to 84x84 (README.md:86-88); EuroSAT: 10 classes
# (datasets/EuroSAT_few_shot.py:85) x 2700 images of 64x64.  The other test sets keep their class counts (cl_list lines of
# datasets/{CropDisease,ISIC,Chest}_few_shot.py) on a 64x64 x 600 stand-in pool.
DATASET_SHAPES = {"miniImageNet": (64, 600, 84), "EuroSAT": (10, 2700, 64), "CropDisease": (38, 600, 64), "ISIC": (7, 600, 64),
                  "ChestX": (7, 600, 64)}


def class_pool_u8(dataset, device, seed=0, n_per_class=None, noise=1.0):
    """A dataset-SHAPED synthetic image pool resident in HBM: uint8 [n_classes, n_per_class, side, side, 3] (what an
    ImageFolder of decoded images is once on the device).  Class c = a low-frequency template + per-image noise, drawn with the
    device generator class by class (a pure function of (seed, dataset, device type)), mapped to pixels as 128 + 48 * x."""
    n_classes, per, side = DATASET_SHAPES[dataset]
    per = int(n_per_class or per)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) * 7919 + 13)
    low = torch.randn((n_classes, 3, 7, 7), generator=g, device=device)
    t = torch.nn.functional.interpolate(low, size=(side, side), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    pool = torch.empty((n_classes, per, side, side, 3), dtype=torch.uint8, device=device)
    for c in range(n_classes):
        x = t[c][None] + noise * torch.randn((per, side, side, 3), generator=g, device=device)
        pool[c] = torch.clamp(torch.round(128.0 + 48.0 * x), 0, 255).to(torch.uint8)
    return pool
