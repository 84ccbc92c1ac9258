"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU *functional* restatement of the reference hot path
(johncai117/Meta-Fine-Tuning): ResNet10 forward / last-block backward, the GNN
few-shot head, the first-order-MAML inner loop and the test-time ``finetune``.
No ``nn.Module`` of the reference is used; every function cites the reference
``file:line`` it restates.  It works on a flat ``dict[str, Tensor]`` keyed with
the reference's state_dict names and is dtype-generic (fp32 and fp64).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``meta-fine-tuning_amd/``) never
does and fails loudly when its HIP library is missing.

Parity pin: the reference has no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, imported on CPU in
the build container by ``oracle/make_golden.py``, ``make_golden_r2.py`` and
``make_golden_g9.py`` (20 fixtures: every §8(a) function incl. the 20-shot and
50-shot configurations, fp32 and fp64 trajectories, 600 + 600 per-episode
accuracies); the resulting vectors live in ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks this file against them everywhere (no
/root/reference needed at test time).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm* default (backbone.py:224, gnn.py:65)
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.01     # F.leaky_relu default (gnn.py:86)

# trunk index -> (indim, outdim, half_res)      backbone.py:417-425, 519-520
STAGES = {4: (64, 64, False), 5: (64, 128, True), 6: (128, 256, True), 7: (256, 512, True)}

# The inner loop adapts the last 9 parameter tensors of ResNet10
# (finetune.py:236-252, gnnnet.py:132-142): trunk.7.{C1,BN1,C2,BN2,shortcut,BNshortcut}
ADAPT_KEYS = [
    "trunk.7.C1.weight", "trunk.7.BN1.weight", "trunk.7.BN1.bias",
    "trunk.7.C2.weight", "trunk.7.BN2.weight", "trunk.7.BN2.bias",
    "trunk.7.shortcut.weight", "trunk.7.BNshortcut.weight", "trunk.7.BNshortcut.bias",
]


# --------------------------------------------------------------------------- batch norm

def batchnorm_train(x, gamma, beta, sd=None, prefix=None, eps=BN_EPS):
    """Train-mode BatchNorm over all dims but channel (dim 1).

    nn.BatchNorm2d/1d forward in training (backbone.py:224-227,409; gnnnet.py:30;
    gnn.py:40,65-74): normalise with the biased batch variance; if ``sd`` has
    running buffers under ``prefix`` update them with momentum 0.1 and the
    *unbiased* variance, and bump ``num_batches_tracked``.
    """
    if sd is not None and prefix is not None and (prefix + ".running_mean") in sd:
        rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
        if rm.dtype == x.dtype:
            out = F.batch_norm(x, rm, rv, gamma, beta, True, BN_MOMENTUM, eps)
        else:                       # fp64 run over fp32 buffers: update copies, write back
            rm2, rv2 = rm.to(x.dtype), rv.to(x.dtype)
            out = F.batch_norm(x, rm2, rv2, gamma, beta, True, BN_MOMENTUM, eps)
            with torch.no_grad():
                rm.copy_(rm2.to(rm.dtype))
                rv.copy_(rv2.to(rv.dtype))
        with torch.no_grad():
            sd[prefix + ".num_batches_tracked"] += 1
        return out
    # same ATen batch-norm kernel the reference's nn.BatchNorm* dispatches to (one fused pass; also what makes the
    # CPU baseline in bench.py a fair stand-in for the reference's own CPU speed)
    return F.batch_norm(x, None, None, gamma, beta, True, BN_MOMENTUM, eps)


def batchnorm_eval(x, gamma, beta, rmean, rvar, eps=BN_EPS):
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - rmean.view(shape)) / torch.sqrt(rvar.view(shape) + eps) * gamma.view(shape) + beta.view(shape)


# --------------------------------------------------------------------------- ResNet10

def simple_block(sd, p, x, idx, train=True, track=True, taps=None):
    """backbone.SimpleBlock.forward (backbone.py:251-261)."""
    indim, outdim, half = STAGES[idx]
    pre = p + "trunk.%d" % idx
    s = 2 if half else 1
    bsd = sd if track else None

    def bn(t, name):
        if train:
            return batchnorm_train(t, sd[pre + name + ".weight"], sd[pre + name + ".bias"], bsd, pre + name)
        return batchnorm_eval(t, sd[pre + name + ".weight"], sd[pre + name + ".bias"],
                              sd[pre + name + ".running_mean"], sd[pre + name + ".running_var"])

    c1 = F.conv2d(x, sd[pre + ".C1.weight"], None, stride=s, padding=1)
    r1 = F.relu(bn(c1, ".BN1"))
    c2 = F.conv2d(r1, sd[pre + ".C2.weight"], None, stride=1, padding=1)
    b2 = bn(c2, ".BN2")
    if indim != outdim:
        sc = F.conv2d(x, sd[pre + ".shortcut.weight"], None, stride=s, padding=0)
        short = bn(sc, ".BNshortcut")
    else:
        sc = None
        short = x
    out = F.relu(b2 + short)
    if taps is not None:
        taps["trunk.%d.C1" % idx] = c1
        taps["trunk.%d.relu1" % idx] = r1
        taps["trunk.%d.C2" % idx] = c2
        if sc is not None:
            taps["trunk.%d.shortcut" % idx] = sc
        taps["trunk.%d.out" % idx] = out
    return out


def resnet10_forward(sd, x, prefix="", train=True, track=True, taps=None):
    """backbone.ResNet.forward with ResNet10(flatten=True) (backbone.py:401-439,519-520).

    The final ``nn.AvgPool2d(7)`` (backbone.py:427) is restated as a global
    average pool: identical at 224x224 (7x7 map) and equal to the
    ``AvgPool2d(3)``-patched reference at 84x84 (3x3 map).
    """
    p = prefix
    bsd = sd if track else None
    c0 = F.conv2d(x, sd[p + "trunk.0.weight"], None, stride=2, padding=3)
    if train:
        b0 = batchnorm_train(c0, sd[p + "trunk.1.weight"], sd[p + "trunk.1.bias"], bsd, p + "trunk.1")
    else:
        b0 = batchnorm_eval(c0, sd[p + "trunk.1.weight"], sd[p + "trunk.1.bias"],
                            sd[p + "trunk.1.running_mean"], sd[p + "trunk.1.running_var"])
    out = F.max_pool2d(F.relu(b0), kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps["trunk.0"] = c0
        taps["trunk.3"] = out
    for idx in (4, 5, 6, 7):
        out = simple_block(sd, p, out, idx, train, track, taps)
    feat = out.mean(dim=(2, 3))
    return feat


# --------------------------------------------------------------------------- Adam

def adam_init(params):
    return {"step": 0, "m": [torch.zeros_like(t) for t in params], "v": [torch.zeros_like(t) for t in params]}


def adam_step(params, grads, state, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam, no amsgrad (finetune.py:255,299; gnnnet.py:128,177; train.py:28).

    Pinned form (torch 1.7.1 ``F.adam``): m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
    denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (lr/(1-b1^t)) * m/denom.
    L2 weight decay (classifier optimisers only) adds wd*p to g first.
    """
    state["step"] += 1
    t = state["step"]
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    step_size = lr / bc1
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, state["m"], state["v"]):
            if weight_decay != 0.0:
                g = g + weight_decay * p
            m.mul_(beta1).add_(g, alpha=1.0 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-step_size)


def sgd_step(params, grads, state, lr=0.01, momentum=0.9, dampening=0.9, weight_decay=0.001):
    """torch.optim.SGD with momentum+dampening+wd (meta_template.py:166; baselinefinetune.py:35).

    First step: buf = g (no dampening); afterwards buf = mom*buf + (1-damp)*g.
    """
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            g = g + weight_decay * p
            if state.get(i) is None:
                state[i] = g.clone()
            else:
                state[i].mul_(momentum).add_(g, alpha=1.0 - dampening)
            p.add_(state[i], alpha=-lr)


# --------------------------------------------------------------------------- inner loop

def inner_step(sd, x_batch, y_batch, adam_state, prefix="", lr=0.01, track=True, return_aux=False):
    """One inner-loop step (finetune.py:276-299; gnnnet.py:156-177).

    ``loss = CrossEntropy(ResNet10(x_batch)[B,512], y)`` on the raw 512-d feature
    used as logits; backward reaches only the last block's 9
    tensors; Adam(lr=0.01) on them.  The backbone is in train mode.
    """
    keys = [prefix + k for k in ADAPT_KEYS]
    params = [sd[k] for k in keys]
    for t in params:
        t.requires_grad_(True)
    taps = {} if return_aux else None
    feat = resnet10_forward(sd, x_batch, prefix, train=True, track=track, taps=taps)
    loss = F.cross_entropy(feat, y_batch)
    grads = torch.autograd.grad(loss, params)
    for t in params:
        t.requires_grad_(False)
    adam_step(params, grads, adam_state, lr=lr)
    if return_aux:
        return loss.detach(), feat.detach(), grads, taps
    return loss.detach()


def clone_state(sd, dtype=None):
    out = OrderedDict()
    for k, v in sd.items():
        if dtype is not None and v.is_floating_point():
            out[k] = v.detach().clone().to(dtype)
        else:
            out[k] = v.detach().clone()
    return out


def feature_state(sd):
    """finetune.py:187-198: keep 'feature.*' keys, strip the prefix."""
    out = OrderedDict()
    for k, v in sd.items():
        if "feature." in k and not k.startswith("feature2.") and not k.startswith("feature3."):
            out[k.replace("feature.", "")] = v.detach().clone()
    return out


# --------------------------------------------------------------------------- GNN head

def wcompute(sd, name, x):
    """gnn.Wcompute.forward, operator 'J2', activation 'softmax' (gnn.py:78-132).

    x: [B, N, F] -> A: [B, N, N] row-softmax affinities with the diagonal masked by
    -1e8 (gnn.py:105-115).  The identity half of the J2 stack is implicit.
    BatchNorm2d(track_running_stats=False): batch statistics over all B*N*N
    positions in train and eval alike (gnn.py:65-74).
    """
    B, N, Fd = x.shape
    d = (x.unsqueeze(2) - x.unsqueeze(1)).abs()          # [B,N,N,F]      gnn.py:79-81
    h = d.reshape(B * N * N, Fd)
    for li in range(1, 5):
        w = sd["%s.conv2d_%d.weight" % (name, li)]
        h = h @ w.view(w.shape[0], -1).t() + sd["%s.conv2d_%d.bias" % (name, li)]
        h = batchnorm_train(h, sd["%s.bn_%d.weight" % (name, li)], sd["%s.bn_%d.bias" % (name, li)])
        h = F.leaky_relu(h, LRELU_SLOPE)
    w = sd[name + ".conv2d_last.weight"]
    s = (h @ w.view(1, -1).t() + sd[name + ".conv2d_last.bias"]).view(B, N, N)
    s = s - torch.eye(N, dtype=x.dtype).unsqueeze(0) * 1e8
    return F.softmax(s, dim=2)


def gconv(sd, name, A, x, bn):
    """gnn.Gconv.forward with gmul, J=2 (gnn.py:16-56): fc(cat(x, A@x)) [+ BN1d over B*N rows]."""
    B, N, Fd = x.shape
    h = torch.cat([x, torch.bmm(A, x)], dim=2).reshape(B * N, 2 * Fd)
    h = h @ sd[name + ".fc.weight"].t() + sd[name + ".fc.bias"]
    if bn:
        h = batchnorm_train(h, sd[name + ".bn.weight"], sd[name + ".bn.bias"])
    return h.view(B, N, -1)


def gnn_forward(sd, nodes, prefix="gnn."):
    """gnn.GNN_nl.forward (gnn.py:154-166): nodes [B,N,128+n_way] -> [B,N,n_way]."""
    x = nodes
    for i in range(2):
        A = wcompute(sd, prefix + "layer_w%d" % i, x)
        xn = F.leaky_relu(gconv(sd, prefix + "layer_l%d" % i, A, x, True), LRELU_SLOPE)
        x = torch.cat([x, xn], dim=2)
    A = wcompute(sd, prefix + "w_comp_last", x)
    return gconv(sd, prefix + "layer_last", A, x, False)


def support_label(n_way, n_support, dtype=torch.float32):
    """GnnNet.__init__ (gnnnet.py:34-38): one-hot rows for supports, zero row per query slot."""
    lab = torch.zeros(n_way, n_support + 1, n_way, dtype=dtype)
    for c in range(n_way):
        lab[c, :n_support, c] = 1.0
    return lab.view(1, n_way * (n_support + 1), n_way)


def gnnnet_scores_from_z(sd, z, n_way, n_support, n_query):
    """z_stack + forward_gnn (gnnnet.py:82-87, 210-217): z [n_way, n_support+n_query, 128] -> [n_way*n_query, n_way]."""
    lab = support_label(n_way, n_support, z.dtype)
    graphs = []
    for i in range(n_query):
        g = torch.cat([z[:, :n_support], z[:, n_support + i:n_support + i + 1]], dim=1).reshape(1, -1, z.shape[2])
        graphs.append(torch.cat([g, lab], dim=2))
    nodes = torch.cat(graphs, dim=0)
    out = gnn_forward(sd, nodes)
    out = out.view(n_query, n_way, n_support + 1, n_way)[:, :, -1].permute(1, 0, 2).contiguous().view(-1, n_way)
    return out


def fc_project(sd, feats):
    """GnnNet.fc = Linear(512,128) + BatchNorm1d(128, no running stats) (gnnnet.py:30)."""
    h = feats @ sd["fc.0.weight"].t() + sd["fc.0.bias"]
    return batchnorm_train(h, sd["fc.1.weight"], sd["fc.1.bias"])


def gnnnet_set_forward(sd, x, n_way, n_support, n_query, is_feature=False, track=True):
    """GnnNet.set_forward (gnnnet.py:68-87)."""
    if is_feature:
        assert x.shape[1] == n_support + 15                 # gnnnet.py:73
        z = fc_project(sd, x.reshape(-1, x.shape[-1]))
    else:
        feats = resnet10_forward(sd, x.reshape(-1, *x.shape[2:]), "feature.", train=True, track=track)
        z = fc_project(sd, feats)
    z = z.view(n_way, -1, z.shape[1])
    return gnnnet_scores_from_z(sd, z, n_way, n_support, n_query)


def fold50_z(z, n_way, ns):
    """gnnnet_copy.py:67-72,232-236: supports k and k+ns averaged -> ns graph nodes per class, queries appended."""
    z3 = z[:, :2 * ns].reshape(n_way, 2, ns, z.shape[-1]).mean(dim=1)
    return torch.cat([z3, z[:, 2 * ns:]], dim=1)


def gnnnet50_set_forward(sd, x, n_way, n_query, is_feature=True, track=True):
    """gnnnet_copy.GnnNet.set_forward (gnnnet_copy.py:52-77): 50 supports folded to 25 by
    averaging support k with support k+25, then the 5-way 25-shot graph (N=130)."""
    ns = 25
    if is_feature:
        assert x.shape[1] == 2 * ns + 15                    # gnnnet_copy.py:56
        feats = x.reshape(-1, x.shape[-1])
    else:
        feats = resnet10_forward(sd, x.reshape(-1, *x.shape[2:]), "feature.", train=True, track=track)
    z = fc_project(sd, feats).view(n_way, -1, 128)
    return gnnnet_scores_from_z(sd, fold50_z(z, n_way, ns), n_way, ns, n_query)


# --------------------------------------------------------------------------- test-time finetune

def finetune_support_set(liz_x, n_way, n_support):
    """finetune.py:208-233: support images of view 0 twice, then of views 1.. ; labels likewise."""
    x = liz_x[0]
    xa = x[:, :n_support].contiguous().view(n_way * n_support, *x.shape[2:])
    ya = torch.from_numpy(np.repeat(np.arange(n_way), n_support))
    xs, ys = [xa, xa], [ya, ya]
    for xv in liz_x[1:]:
        xs.append(xv[:, :n_support].contiguous().view(n_way * n_support, *x.shape[2:]))
        ys.append(ya)
    return torch.cat(xs, 0), torch.cat(ys, 0)


def finetune_perms(n_total, total_epoch, rng=np.random):
    """finetune.py:270-272: one ``np.random.permutation`` per epoch from the global numpy RNG."""
    return [rng.permutation(n_total) for _ in range(total_epoch)]


def finetune_episode(state, liz_x, n_way=5, n_support=5, total_epoch=5, perms=None, batch_size=5,
                     dtype=torch.float32, return_feats=False, dead_query_pass=False, fold50=False):
    """finetune.finetune (finetune.py:182-328), method gnnnet, flatten=True, freeze_backbone=False.
    ``fold50``: finetune_50.finetune (finetune_50.py:182-330) -- the same loop, scored by gnnnet_copy.GnnNet.set_forward
    (gnnnet_copy.py:52-77; ``n_support`` is then the TRUE support count, 50).

    state: full GnnNet state dict (feature.*, fc.*, gnn.*).  Returns softmax scores
    [n_way*n_query, n_way].  ``perms`` defaults to draws from the global numpy RNG
    in reference order.  ``dead_query_pass`` replays finetune.py:307, whose output is
    unused (it only moves BN running stats that nothing reads).
    """
    sd_all = clone_state(state, dtype)
    fsd = feature_state(sd_all)
    x0 = liz_x[0].to(dtype)
    n_query = x0.shape[1] - n_support
    xa, ya = finetune_support_set([v.to(dtype) for v in liz_x], n_way, n_support)
    n_total = xa.shape[0]
    assert n_total == n_way * n_support * (len(liz_x) + 1)     # finetune.py:269 'lengt'
    if perms is None:
        perms = finetune_perms(n_total, total_epoch)
    adam = adam_init([fsd[k] for k in ADAPT_KEYS])
    for ep in range(total_epoch):
        rand_id = perms[ep]
        for j in range(0, n_total, batch_size):
            sel = torch.from_numpy(np.asarray(rand_id[j:min(j + batch_size, n_total)]))
            inner_step(fsd, xa[sel], ya[sel], adam)
    x_inn = x0.reshape(n_way * (n_support + n_query), *x0.shape[2:])
    with torch.no_grad():
        feats = resnet10_forward(fsd, x_inn, "", train=True).view(n_way, n_support + n_query, -1)
        if dead_query_pass:
            xb = x0[:, n_support:].contiguous().view(n_way * n_query, *x0.shape[2:])
            resnet10_forward(fsd, xb, "", train=True)
        if fold50:
            scores = gnnnet50_set_forward(sd_all, feats, n_way, n_query, is_feature=True)
        else:
            scores = gnnnet_set_forward(sd_all, feats, n_way, n_support, n_query, is_feature=True)
        out = F.softmax(scores, dim=1)
    if return_feats:
        return out, feats, fsd
    return out


def finetune_frozen_episode(state, liz_x, n_way=5, n_support=5, total_epoch=5, dtype=torch.float32):
    """finetune.finetune(..., freeze_backbone=True) (finetune.py:253-266,270-299,306-317): the backbone is in eval mode
    and has no optimiser, the classifier never receives a gradient -- the "fine-tuning" loop only draws its
    permutations (one per epoch, consumed here for stream parity) -- so the scores are the GNN on eval-mode features."""
    sd_all = clone_state(state, dtype)
    fsd = feature_state(sd_all)
    x0 = liz_x[0].to(dtype)
    n_query = x0.shape[1] - n_support
    for _ in range(total_epoch):
        np.random.permutation(n_way * n_support * (len(liz_x) + 1))
    with torch.no_grad():
        feats = resnet10_forward(fsd, x0.reshape(n_way * (n_support + n_query), *x0.shape[2:]), "", train=False)
        scores = gnnnet_set_forward(sd_all, feats.view(n_way, n_support + n_query, -1), n_way, n_support, n_query, is_feature=True)
        return F.softmax(scores, dim=1)


def finetune_linear_episode(state, liz_x, n_way=5, n_support=5, w0=None, b0=None, perms=None, epochs=20,
                            batch_size=5, dtype=torch.float32):
    """finetune.finetune_linear (finetune.py:45-174), freeze_backbone=False: a Linear(512, n_way) classifier
    (initial weight/bias ``w0``/``b0``; the reference draws them from torch's RNG, finetune.py:65) on top of the
    backbone; 20 epochs over the n_way*n_support ORIGINAL support images only (finetune.py:139-141 permutes
    ``support_size``, so the augmented views appended at :93-101 are never drawn), mini-batches of 5;
    Adam(lr .01, weight_decay .001) on the classifier, Adam(lr .01) on the last ResNet block (:107-124).
    Scores = softmax(classifier(backbone(cat(support, query))[n_support*n_way:])) (:165-174)."""
    sd_all = clone_state(state, dtype)
    fsd = feature_state(sd_all)
    x0 = liz_x[0].to(dtype)
    n_query = x0.shape[1] - n_support
    xa = x0[:, :n_support].contiguous().view(n_way * n_support, *x0.shape[2:])
    xb = x0[:, n_support:].contiguous().view(n_way * n_query, *x0.shape[2:])
    ya = torch.from_numpy(np.repeat(np.arange(n_way), n_support))
    support_size = n_way * n_support
    if perms is None:
        perms = [np.random.permutation(support_size) for _ in range(epochs)]
    w = w0.detach().clone().to(dtype)
    b = b0.detach().clone().to(dtype)
    params = [fsd[k] for k in ADAPT_KEYS]
    adam_blk = adam_init(params)
    adam_cls = adam_init([w, b])
    for ep in range(epochs):
        rand_id = perms[ep]
        for j in range(0, support_size, batch_size):
            sel = torch.from_numpy(np.asarray(rand_id[j:min(j + batch_size, support_size)]))
            for t in params + [w, b]:
                t.requires_grad_(True)
            feat = resnet10_forward(fsd, xa[sel], "", train=True)
            loss = F.cross_entropy(F.linear(feat, w, b), ya[sel])
            grads = torch.autograd.grad(loss, params + [w, b])
            for t in params + [w, b]:
                t.requires_grad_(False)
            adam_step([w, b], list(grads[-2:]), adam_cls, lr=0.01, weight_decay=0.001)
            adam_step(params, list(grads[:-2]), adam_blk, lr=0.01)
    with torch.no_grad():
        out = resnet10_forward(fsd, torch.cat([xa, xb], 0), "", train=True)[support_size:]
        return F.softmax(F.linear(out, w, b), dim=1)


def finetune_linear_frozen_episode(state, liz_x, n_way=5, n_support=5, w0=None, b0=None, perms=None, epochs=20, batch_size=5,
                                   dtype=torch.float32):
    """finetune.finetune_linear(..., freeze_backbone=True) (finetune.py:45-174 with :123-135,:144,:163 taking the frozen
    branch): the backbone is in eval mode and has no optimiser, so its features are constants; only the Linear(512, n_way)
    classifier is trained -- Adam(lr .01, weight_decay .001), 20 epochs over the n_way*n_support original support images in
    mini-batches of 5 -- and the scores are softmax(classifier(backbone_eval(query)))."""
    sd_all = clone_state(state, dtype)
    fsd = feature_state(sd_all)
    x0 = liz_x[0].to(dtype)
    n_query = x0.shape[1] - n_support
    xa = x0[:, :n_support].contiguous().view(n_way * n_support, *x0.shape[2:])
    xb = x0[:, n_support:].contiguous().view(n_way * n_query, *x0.shape[2:])
    ya = torch.from_numpy(np.repeat(np.arange(n_way), n_support))
    support_size = n_way * n_support
    if perms is None:
        perms = [np.random.permutation(support_size) for _ in range(epochs)]
    w = w0.detach().clone().to(dtype)
    b = b0.detach().clone().to(dtype)
    adam_cls = adam_init([w, b])
    with torch.no_grad():
        za = resnet10_forward(fsd, xa, "", train=False)
        zb = resnet10_forward(fsd, xb, "", train=False)
    for ep in range(epochs):
        rand_id = perms[ep]
        for j in range(0, support_size, batch_size):
            sel = torch.from_numpy(np.asarray(rand_id[j:min(j + batch_size, support_size)]))
            for t in (w, b):
                t.requires_grad_(True)
            loss = F.cross_entropy(F.linear(za[sel], w, b), ya[sel])
            grads = torch.autograd.grad(loss, [w, b])
            for t in (w, b):
                t.requires_grad_(False)
            adam_step([w, b], list(grads), adam_cls, lr=0.01, weight_decay=0.001)
    with torch.no_grad():
        return F.softmax(F.linear(zb, w, b), dim=1)


# --------------------------------------------------------------------------- meta-train / meta-fine-tune

def meta_train_loss(sd, x, n_way, n_support, track=True, fold50=False):
    """GnnNet.set_forward_loss (gnnnet.py:219-224): CE(scores, repeat(range(n_way), n_query)).
    ``fold50``: gnnnet_copy.GnnNet.set_forward_loss (gnnnet_copy.py:258-263; n_support = the true 50)."""
    n_query = x.shape[1] - n_support
    if fold50:
        scores = gnnnet50_set_forward(sd, x, n_way, n_query, is_feature=False, track=track)
    else:
        scores = gnnnet_set_forward(sd, x, n_way, n_support, n_query, is_feature=False, track=track)
    y = torch.from_numpy(np.repeat(np.arange(n_way), n_query))
    return F.cross_entropy(scores, y), scores


def maml_update(feature, feature2, feature3, prefix="feature."):
    """GnnNet.MAML_update (gnnnet.py:90-103): theta <- theta - (theta3 - theta2) on the last 9 tensors."""
    with torch.no_grad():
        for k in ADAPT_KEYS:
            feature[prefix + k].sub_(feature3[k] - feature2[k])


def set_forward_finetune(sd, x, n_way, n_support, mem, perms=None, total_epoch=15, batch_size=4, fold50=False):
    """GnnNet.set_forward_finetune (gnnnet.py:106-208).  ``mem`` carries first/feature2/feature3
    between calls.  Returns scores [n_way*16, n_way] with autograd attached to ``sd`` tensors.
    ``fold50``: gnnnet_copy.GnnNet.set_forward_finetune (gnnnet_copy.py:135-246): ``n_support`` = the true support count
    (50), the caller passes total_epoch=5 (:177), and the supports are pair-averaged before the graph (:232-236)."""
    n_query = x.shape[1] - n_support
    if not mem.get("first", True):
        maml_update(sd, mem["feature2"], mem["feature3"])               # gnnnet.py:122
    xb = x[:, n_support:].contiguous().view(n_way * n_query, *x.shape[2:])
    xa = x[:, :n_support].contiguous().view(n_way * n_support, *x.shape[2:])
    ya = torch.from_numpy(np.repeat(np.arange(n_way), n_support))
    fsd = feature_state(OrderedDict((k, v) for k, v in sd.items()))      # deepcopy(self.feature)  :126
    support_size = n_way * n_support
    if perms is None:
        perms = finetune_perms(support_size, total_epoch)
    adam = adam_init([fsd[k] for k in ADAPT_KEYS])
    for ep in range(total_epoch):
        rand_id = perms[ep]
        for j in range(0, support_size, batch_size):
            sel = torch.from_numpy(np.asarray(rand_id[j:min(j + batch_size, support_size)]))
            inner_step(fsd, xa[sel], ya[sel], adam)
    mem["first"] = False
    mem["feature2"] = feature_state(OrderedDict((k, v) for k, v in sd.items()))   # :185
    mem["feature3"] = clone_state(fsd)                                            # :186
    with torch.no_grad():                                                         # load_state_dict :187
        for k, v in fsd.items():
            sd["feature." + k].copy_(v)
    fs = resnet10_forward(sd, xa, "feature.", train=True).view(n_way, n_support, -1)   # :192
    fq = resnet10_forward(sd, xb, "feature.", train=True).view(n_way, n_query, -1)     # :193
    final = torch.cat([fs, fq], dim=1)
    assert final.shape[1] == n_support + 16                                           # :198
    z = fc_project(sd, final.view(-1, final.shape[-1])).view(n_way, -1, 128)
    if fold50:
        return gnnnet_scores_from_z(sd, fold50_z(z, n_way, n_support // 2), n_way, n_support // 2, n_query)
    return gnnnet_scores_from_z(sd, z, n_way, n_support, n_query)


# --------------------------------------------------------------------------- linear-head adaptation

def set_forward_adaptation(z_all, n_way, n_support, w0, b0, perms=None, epochs=100, batch_size=4):
    """MetaTemplate/BaselineFinetune.set_forward_adaptation (meta_template.py:153-186,
    baselinefinetune.py:17-58): SGD(lr .01, mom .9, damp .9, wd 1e-3) on a fresh Linear(512,n_way).
    ``w0``/``b0`` are the initial head parameters (the reference draws them from torch's RNG)."""
    n_query = z_all.shape[1] - n_support
    zs = z_all[:, :n_support].contiguous().view(n_way * n_support, -1)
    zq = z_all[:, n_support:].contiguous().view(n_way * n_query, -1)
    ys = torch.from_numpy(np.repeat(np.arange(n_way), n_support))
    w, b = w0.clone(), b0.clone()
    st = {}
    support_size = n_way * n_support
    if perms is None:
        perms = finetune_perms(support_size, epochs)
    for ep in range(epochs):
        rid = perms[ep]
        for i in range(0, support_size, batch_size):
            sel = torch.from_numpy(np.asarray(rid[i:min(i + batch_size, support_size)]))
            w.requires_grad_(True)
            b.requires_grad_(True)
            loss = F.cross_entropy(zs[sel] @ w.t() + b, ys[sel])
            gw, gb = torch.autograd.grad(loss, [w, b])
            w = w.detach()
            b = b.detach()
            sgd_step([w, b], [gw, gb], st)
    return zq @ w.t() + b
