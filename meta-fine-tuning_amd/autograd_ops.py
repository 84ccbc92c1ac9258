"""Bridge between the reference-shaped nn.Modules (OIHW nn.Parameters, autograd, optimisers) and the
HIP functional layer.  Parameters are repacked into the kernels' layouts lazily (cache keyed on the
tensors' autograd version counters) and gradients are handed back in the reference's layouts.
"""
import torch

from . import functional as Fn
from . import functional_bwd as FB
from . import ops

_ARENAS = {}


def arena_for(device):
    a = _ARENAS.get(device)
    if a is None:
        a = Fn.Arena(device)
        _ARENAS[device] = a
    return a


def _require_cuda(x, what):
    if not x.is_cuda:
        raise RuntimeError("%s: input is on %s -- the MI355X path has no CPU fallback; call .cuda() first" % (what, x.device))


def module_weights(mod):
    """Packed ResNet10Weights of a backbone.ResNet module, rebuilt only when a parameter changed."""
    params = list(mod.parameters())
    key = tuple((p.data_ptr(), p._version) for p in params)
    cache = mod.__dict__.get("_mft_pack")
    if cache is None or cache[0] != key:
        if cache is not None and cache[1].can_repack() and _same_storage(cache[0], key):
            cache[1].repack()             # same tensors, new values (optimizer.step): one launch refreshes every packed copy
            W = cache[1]
        else:
            sd = {k: v for k, v in mod.state_dict().items()}
            W = Fn.ResNet10Weights(sd, params[0].device)
        cache = (key, W)
        mod.__dict__["_mft_pack"] = cache
    return cache[1]


def _same_storage(old_key, new_key):
    return len(old_key) == len(new_key) and all(a[0] == b[0] for a, b in zip(old_key, new_key))


def _running(mod):
    """name -> (running_mean, running_var, num_batches_tracked) of every tracking BatchNorm2d: the statistics launch of each layer
    advances all three (mft_bn_stats), so a forward costs no counter launch of its own."""
    run = {}
    for name, m in mod.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d) and m.track_running_stats:
            run[name] = (m.running_mean, m.running_var, m.num_batches_tracked)
    return run


def _eval_weights(mod, W):
    """Eval-mode BatchNorm: statistics come from the running buffers."""
    stats = {}
    lib = ops._lib.lib()
    for name, m in mod.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            rstd = torch.empty_like(m.running_var)
            ops._lib.check(lib.mft_var_to_rstd(ops._p(m.running_var), ops._p(rstd), rstd.numel(), m.eps, ops._stream()),
                           "mft_var_to_rstd")
            stats[name] = (m.running_mean.view(1, -1), rstd.view(1, -1))
    return stats


class _ResNet10Fn(torch.autograd.Function):
    """ResNet10 forward on HIP; backward through the last block only (the inner-loop regime: everything
    below trunk.7 is frozen -- finetune.py:242-252)."""

    @staticmethod
    def forward(ctx, mod, x_nhwc, *params):
        W = module_weights(mod)
        arena = arena_for(x_nhwc.device)
        n = x_nhwc.shape[0]
        slab = Fn.LastBlockSlab(1, x_nhwc.device, zero=False)
        slab.load_shared(W)
        tape = {}
        feat = Fn.resnet10_forward(W, x_nhwc, arena, ipg=n, slab=slab, tape=tape, running=_running(mod), tag="mod%d" % n)
        ctx.tape, ctx.slab, ctx.n, ctx.mod = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tape.items()}, slab, n, mod
        return feat.clone()

    @staticmethod
    def backward(ctx, dfeat):
        mod = ctx.mod
        names = [n for n, _ in mod.named_parameters()]
        need = [p.requires_grad for p in mod.parameters()]
        if any(need[:-9]):
            raise NotImplementedError(
                "backward below trunk.7 requested (%s): use methods.gnnnet.GnnNet.set_forward_loss for the "
                "meta-train path" % [n for n, r in zip(names, need) if r][0])
        grads = Fn.LastBlockSlab(1, dfeat.device)
        arena = arena_for(dfeat.device)
        Fn.last_block_backward(ctx.tape, dfeat.contiguous(), ctx.slab, grads, arena, ipg=ctx.n, tag="modbw%d" % ctx.n)
        g = grads.export(0)
        out = [None] * len(names)
        for i, nm in enumerate(names):
            if nm in g and need[i]:
                out[i] = g[nm]
        return (None, None) + tuple(out)


class _ResNet10FullFn(torch.autograd.Function):
    """ResNet10 forward + full backward on HIP (meta-training: every parameter receives a gradient,
    train.py:28; meta_template.py:76-92)."""

    @staticmethod
    def forward(ctx, mod, x_nhwc, groups, *params):
        W = module_weights(mod)
        feat, tape = FB.resnet10_forward_taped(W, x_nhwc, running=_running(mod), groups=groups)
        ctx.tape, ctx.W, ctx.mod = tape, W, mod
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        mod = ctx.mod
        names = [n for n, _ in mod.named_parameters()]
        need = [p.requires_grad for p in mod.parameters()]
        g = FB.resnet10_backward(ctx.W, ctx.tape, dfeat.contiguous().float(), set(names))
        out = [g[nm] if r else None for nm, r in zip(names, need)]
        return (None, None, None) + tuple(out)


class _HeadFn(torch.autograd.Function):
    """GnnNet.fc + graph assembly + GNN_nl + score gather with a hand-written backward (gnnnet.py:76-87,210-217)."""

    @staticmethod
    def forward(ctx, model, feats, n_support, n_query, fold, episodes, *params):
        G = head_weights(model.gnn, model.fc, model.n_way)
        scores, tape = FB.head_forward_taped(G, feats.contiguous().float(), model.n_way, n_support, n_query, fold, episodes)
        ctx.tape, ctx.G, ctx.model = tape, G, model
        ctx.feats_need = feats.requires_grad
        return scores

    @staticmethod
    def backward(ctx, dscores):
        model = ctx.model
        dfeats, g = FB.head_backward(ctx.G, ctx.tape, dscores.contiguous().float())
        names = ["fc." + n for n, _ in model.fc.named_parameters()] + ["gnn." + n for n, _ in model.gnn.named_parameters()]
        plist = list(model.fc.parameters()) + list(model.gnn.parameters())
        out = [g[nm].view_as(p) if p.requires_grad else None for nm, p in zip(names, plist)]
        return (None, dfeats if ctx.feats_need else None, None, None, None, None) + tuple(out)


class _CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(scores, y) = mean_r(logsumexp(scores_r) - scores_r[y_r]) with ONE launch each way
    (mft_cross_entropy_mean / _backward; gnnnet.py:219-231, baselinetrain.py:38-45).  Labels are read as given (int64 or int32)."""

    @staticmethod
    def forward(ctx, logits, target, loss_sum):
        rows, C = logits.shape
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        lib = ops._lib.lib()
        ops._lib.check(lib.mft_cross_entropy_mean(ops._p(logits), logits.stride(0), ops._p(target), 1 if target.dtype == torch.int64 else 0,
                                                  C, rows, ops._p(loss), ops._p(loss_sum), ops._stream()), "mft_cross_entropy_mean")
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        rows, C = logits.shape
        d = torch.empty((rows, C), device=logits.device, dtype=torch.float32)
        g = g.contiguous().float()
        lib = ops._lib.lib()
        ops._lib.check(lib.mft_cross_entropy_mean_backward(ops._p(logits), logits.stride(0), ops._p(target),
                                                           1 if target.dtype == torch.int64 else 0, C, rows, ops._p(g), ops._p(d), C,
                                                           ops._stream()), "mft_cross_entropy_mean_backward")
        return d, None, None


class CrossEntropyLoss(torch.nn.CrossEntropyLoss):
    """``nn.CrossEntropyLoss()`` as the episode / pre-training losses use it (gnnnet.py:43, baselinetrain.py:20: default arguments,
    [rows, C] float scores against [rows] class indices), computed by the HIP library.  Same class hierarchy and attributes as
    torch's module; anything outside that use (class weights, label smoothing, probabilities as targets, a CPU tensor) raises --
    there is no torch fallback on the product path."""

    def forward(self, input, target):
        if (self.weight is not None or self.reduction != "mean" or self.label_smoothing != 0.0 or input.dim() != 2 or target.dim() != 1
                or target.dtype not in (torch.int64, torch.int32) or target.shape[0] != input.shape[0]):
            raise NotImplementedError("CrossEntropyLoss on the HIP path: default options, [rows, C] scores, [rows] int64 / int32 labels")
        _require_cuda(input, "CrossEntropyLoss")
        if not target.is_cuda:
            raise RuntimeError("CrossEntropyLoss: labels are on the CPU -- the MI355X path has no CPU fallback; call .cuda() first")
        if input.dtype != torch.float32 or input.stride(1) != 1:
            input = input.float().contiguous()
        return _CrossEntropyFn.apply(input, target.contiguous(), self.loss_sum(input.device))

    def loss_sum(self, device=None):
        """float64 device scalar that every forward of this module adds its loss to (one per device, never re-created: a recorded
        hipGraph keeps writing it).  The episode loops print running means from differences of it -- one read-back per printed
        line instead of one per step (meta_template.py:91-93)."""
        sums = self.__dict__.setdefault("_mft_loss_sums", {})
        if device is None:
            return next(iter(sums.values()), None)
        key = torch.device(device).index
        t = sums.get(key)
        if t is None:
            t = sums[key] = torch.zeros((), device=device, dtype=torch.float64)
        return t


def resnet10_module_forward(mod, x, groups=1):
    """backbone.ResNet.forward: x NCHW [n,3,H,W] on the GPU -> [n,512].  ``groups`` > 1 (meta-training only): the n images are
    ``groups`` episodes one after the other, each a BatchNorm mini-batch of its own (GnnNet.set_forward_loss_lockstep)."""
    _require_cuda(x, "ResNet10.forward")
    if groups != 1 and not (mod.training and torch.is_grad_enabled() and all(p.requires_grad for p in mod.parameters())):
        raise NotImplementedError("several BatchNorm groups per call exist on the meta-training path only (train mode, every "
                                  "backbone parameter trainable)")
    if x.dim() == 4 and x.dtype == torch.float32 and x.permute(0, 2, 3, 1).is_contiguous():
        xn = x.permute(0, 2, 3, 1)            # already NHWC in memory (train.ResidentEpisodeLoader: mft_augment_views writes NHWC)
    else:
        x = x.contiguous().float()
        xn = ops.nchw_to_nhwc(x)
    params = list(mod.parameters())
    if not mod.training:
        # eval-mode BatchNorm (finetune(freeze_backbone=True), finetune.py:265-266): running statistics, no buffer updates
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            raise NotImplementedError("autograd through an eval-mode backbone is not on the HIP hot path (the reference only "
                                      "evaluates it: the frozen-backbone loop has no optimiser, finetune.py:253-299)")
        W = module_weights(mod)
        n = xn.shape[0]
        return Fn.resnet10_forward(W, xn, arena_for(xn.device), ipg=n, fixed=_eval_weights(mod, W), tag="evl%d" % n).clone()
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if any(p.requires_grad for p in params[:-9]):
            out = _ResNet10FullFn.apply(mod, xn, groups, *params)       # meta-training: gradients for the whole backbone
        else:
            out = _ResNet10Fn.apply(mod, xn, *params)           # inner loop: last block only
    else:
        W = module_weights(mod)
        n = xn.shape[0]
        out = Fn.resnet10_forward(W, xn, arena_for(xn.device), ipg=n, running=_running(mod), tag="mod%d" % n).clone()
    return out            # (every BatchNorm's num_batches_tracked was advanced by its own statistics launch: _running)


# ---------------------------------------------------------------------------------------------- packing

def pad_rows(x2d, ld):
    out = torch.zeros((x2d.shape[0], ld), device=x2d.device)
    out[:, :x2d.shape[1]] = x2d
    return out


def _version_key(mod):
    return tuple((p.data_ptr(), p._version) for p in mod.parameters())


def head_weights(gnn_mod, fc_mod=None, n_way=None):
    """Packed GnnHeadWeights for a GNN_nl (and optionally GnnNet.fc), cached on the parameter versions."""
    key = _version_key(gnn_mod) + (() if fc_mod is None else _version_key(fc_mod))
    cache = gnn_mod.__dict__.get("_mft_pack")
    if cache is not None and cache[0] != key and fc_mod is not None and cache[1].can_repack() and _same_storage(cache[0], key):
        cache[1].repack()
        cache = (key, cache[1])
        gnn_mod.__dict__["_mft_pack"] = cache
    if cache is None or cache[0] != key:
        sd = {"gnn." + k: v for k, v in gnn_mod.state_dict().items()}
        if fc_mod is not None:
            sd.update({"fc." + k: v for k, v in fc_mod.state_dict().items()})
        else:
            dev = next(gnn_mod.parameters()).device
            sd.update({"fc.0.weight": torch.zeros(128, 512, device=dev), "fc.0.bias": torch.zeros(128, device=dev),
                       "fc.1.weight": torch.ones(128, device=dev), "fc.1.bias": torch.zeros(128, device=dev)})
        G = Fn.GnnHeadWeights(sd, next(gnn_mod.parameters()).device, gnn_mod.train_N_way if n_way is None else n_way)
        cache = (key, G)
        gnn_mod.__dict__["_mft_pack"] = cache
    return cache[1]


class _Solo:
    """GnnHeadWeights-shaped view of a single Wcompute / Gconv module (stand-alone module calls)."""

    def __init__(self):
        self.wc, self.gc, self.n_way = {}, {}, 0


def solo_weights(mod, kind):
    key = _version_key(mod)
    cache = mod.__dict__.get("_mft_pack")
    if cache is None or cache[0] != key:
        dev = next(mod.parameters()).device
        S = _Solo()

        def d(t):
            return t.detach().to(dev).float().contiguous()
        if kind == "wc":
            layers = []
            for li in range(1, 5):
                conv, bn = getattr(mod, "conv2d_%d" % li), getattr(mod, "bn_%d" % li)
                layers.append((ops.pack_conv_weight(d(conv.weight)), d(conv.bias), d(bn.weight), d(bn.bias), conv.weight.shape[0]))
            S.wc["solo"] = (layers, (ops.pack_conv_weight(d(mod.conv2d_last.weight)), d(mod.conv2d_last.bias)))
        else:
            g = b = None
            if mod.bn_bool:
                g, b = d(mod.bn.weight), d(mod.bn.bias)
            S.gc["solo"] = (ops.pack_conv_weight(d(mod.fc.weight)), d(mod.fc.bias), g, b, mod.fc.weight.shape[0])
        cache = (key, S)
        mod.__dict__["_mft_pack"] = cache
    return cache[1]


def invalidate(mod):
    """Drop cached packed weights after an in-place parameter update done behind autograd's back."""
    for m in mod.modules():
        m.__dict__.pop("_mft_pack", None)


def touch(mod):
    """Mark the cached packed weights stale but keep them: the next lookup refreshes every packed copy IN PLACE (one launch) instead
    of rebuilding the pack -- for in-place parameter updates that do not bump autograd's version counters (kernels writing
    ``p.data``).  Anything recorded against the pack's buffers (a captured inner loop) stays valid."""
    for m in mod.modules():
        c = m.__dict__.get("_mft_pack")
        if c is not None:
            m.__dict__["_mft_pack"] = (tuple((k[0], -1) for k in c[0]), c[1])


def gnnnet_head(model, feats, n_support, n_query, fold=False, episodes=1):
    """GnnNet.fc + z_stack + forward_gnn (gnnnet.py:76-87,210-217): feats [episodes*n_way*(S+n_query), 512] (one episode after
    the other) -> scores [episodes*n_way*n_query, n_way].  Differentiable (hand-written backward) when autograd is recording."""
    _require_cuda(feats, "GnnNet head")
    plist = list(model.fc.parameters()) + list(model.gnn.parameters())
    if torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in plist)):
        return _HeadFn.apply(model, feats, n_support, n_query, fold, episodes, *plist)
    G = head_weights(model.gnn, model.fc, model.n_way)
    f = feats.detach().contiguous().float()
    return Fn.gnnnet_scores(G, f, episodes, model.n_way, n_support, n_query, arena_for(f.device), fold=fold, tag="head%d" % episodes).clone()
