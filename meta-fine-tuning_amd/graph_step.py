"""One meta-training episode's ``loss_fn(x)`` + ``loss.backward()`` as ONE hipGraph replay.

The reference's loop body (meta_template.py:76-92: zero_grad, set_forward_loss, backward, optimizer.step) is ~240 C-ABI launches
(~440 kernels) on ONE 105-image episode: most kernels run for 5-50 us and the step is bound by launch gaps, not by the GPU.
After ``warmup`` eager steps (real training steps, on a side stream as graph capture requires) the forward + backward of the
step is captured once and replayed per episode on a static input buffer; the optimizer step -- and, under episode-parallel
training, the gradient all-reduce in front of it -- stay outside the graph.  Results are bit-identical to the eager loop
(tests/test_metatrain_gpu.py); measured 6.35 -> 4.82 ms per step (profiles/r03_e_metatrain_graph.txt).

Not a tracing compiler: the same hand-written launches, recorded by the HIP runtime on the stream they are issued on."""
import os
import warnings

import torch

from . import settings

ENABLED = settings.current().train_graph


def bump_versions(tensors):
    """Advance autograd's version counter of every tensor by one WITHOUT a device launch: parameters were (or are about to be)
    written through raw pointers, and the weight-pack caches of autograd_ops key on ``p._version``."""
    tensors = tuple(tensors)
    if not tensors:
        return
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        setter(tensors, tuple(t._version + 1 for t in tensors))
    else:                                    # (a torch without the setter: one fused no-op write bumps the counters, at one launch)
        with torch.no_grad():
            torch._foreach_add_(list(tensors), 0)


class GraphedLossBackward:
    """``loss = step(x, optimizer)`` leaves the loss in a static tensor and the gradients in ``p.grad`` exactly as
    ``optimizer.zero_grad(); loss = loss_fn(x); loss.backward()`` would.  The caller must not drop the gradients between
    steps (no ``zero_grad(set_to_none=True)``): the replay overwrites them in place.

    Everything the recording bakes in is part of the replay key: input shape / dtype, n_way / n_query, every parameter's address
    AND requires_grad flag, every sub-module's train / eval mode and BatchNorm momentum.  A freeze / unfreeze, an ``eval()`` call
    or a changed momentum between two calls therefore starts over (eager warm-up steps, new recording) instead of replaying a
    graph of the old configuration; the trainable list is rebuilt with it.  Parameters the optimizer owns but this step does
    not produce a gradient for get ``grad = None`` every step, as ``optimizer.zero_grad()`` in the eager loop gives them."""

    def __init__(self, model, loss_fn, warmup=3):
        self.model, self.loss_fn, self.warmup = model, loss_fn, warmup
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._opt_id, self._opt_others = None, ()
        self.seen = 0
        self.graph = None
        self.key = None
        self.static_x = self.static_loss = None
        self.side = None
        self.failed = False
        self.rekeyed = 0
        self._walk = None

    def _key(self, x):
        # the module tree is walked ONCE (the model's structure does not change between steps): per step the key costs only the
        # address / flag reads -- this is the replay path, which exists to remove host work
        if self._walk is None:
            self._walk = (list(self.model.parameters()), list(self.model.modules()))
        ps, ms = self._walk
        mods = tuple((m.training, getattr(m, "momentum", None)) for m in ms)
        return (tuple(x.shape), x.dtype, self.model.n_way, self.model.n_query, tuple((p.data_ptr(), p.requires_grad) for p in ps), mods)

    def _clear_foreign_grads(self, optimizer):
        """What ``optimizer.zero_grad()`` does for the parameters this step does not differentiate (frozen, or not the model's):
        a stale ``.grad`` must not keep feeding ``optimizer.step()``."""
        if optimizer is None:
            return
        ident = (id(optimizer), self.key)
        if ident != self._opt_id:
            mine = {id(p) for p in self.params}
            self._opt_others = tuple(p for g in optimizer.param_groups for p in g["params"] if id(p) not in mine)
            self._opt_id = ident
        for p in self._opt_others:
            p.grad = None

    def _eager(self, x, stream=None):
        for p in self.params:
            p.grad = None
        if stream is None:
            loss = self.loss_fn(x)
            loss.backward()
            return loss
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            loss = self.loss_fn(x)
            loss.backward()
        torch.cuda.current_stream().wait_stream(stream)
        return loss

    def _capture(self, x):
        from . import autograd_ops as AG  # noqa: F401  (packed weight copies are refreshed inside the captured forward)
        self.static_x = x.detach().to("cuda", copy=True)
        self._one = torch.ones((), device=self.static_x.device, dtype=torch.float32)
        bump_versions(self.params)                      # the captured forward must contain the repack launches
        for p in self.params:
            p.grad = None
        g = torch.cuda.CUDAGraph()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with torch.cuda.graph(g):
                self.static_loss = self.loss_fn(self.static_x)
                self.static_loss.backward(self._one)     # (autograd's implicit ones_like(loss) would be a fill launch in every replay)
        self.graph = g

    def __call__(self, x, optimizer=None):
        key = self._key(x)
        if key != self.key:                         # new shape / parameter tensors / requires_grad flags / train-eval modes: start over
            if self.graph is not None:
                self.rekeyed += 1
                if self.rekeyed > 4:                # a loader that keeps changing the episode shape: recording costs more than it saves
                    self.failed = True
            self.key, self.seen, self.graph = key, 0, None
            self.params = [p for p in self.model.parameters() if p.requires_grad]
        self._clear_foreign_grads(optimizer)
        if self.failed:
            return self._eager(x)
        if self.graph is None:
            if self.seen < self.warmup:
                self.seen += 1
                if self.side is None:
                    self.side = torch.cuda.Stream()
                return self._eager(x.cuda() if not x.is_cuda else x, self.side)
            try:
                self._capture(x)
            except Exception as e:   # noqa: BLE001 -- anything the runtime refuses to record: stay on the eager loop
                warnings.warn("hipGraph capture of the meta-training step failed (%s: %s); continuing without it" % (type(e).__name__, e))
                self.failed, self.graph = True, None
                torch.cuda.synchronize()
                return self._eager(x)
        self.static_x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_loss


class _PreparedStep:
    """Meta-fine-tuning episode (gnnnet.py:106-231): the host-driven half (MAML_update, inner loop, theta_pre / theta_adapted)
    runs eagerly -- its inner loop is a graph of its own (engine.adapt_last_block) -- then the differentiable half (two backbone
    forwards, fc + GNN, loss, backward) is replayed from a hipGraph."""

    def __init__(self, model):
        self.model = model
        self.inner = GraphedLossBackward(model, model.set_forward_loss_finetune_prepared)

    @property
    def graph(self):
        return self.inner.graph

    @property
    def failed(self):
        return self.inner.failed

    def __call__(self, x, optimizer=None):
        x = x.cuda() if not x.is_cuda else x
        self.model._finetune_prepare(x)
        return self.inner(x, optimizer)


def for_loop(model, loss_fn):
    """The graphed step for ``model``'s episode loop, or None when it cannot be replayed.  ``set_forward_loss``: forward + backward
    in one graph; ``set_forward_loss_finetune``: the differentiable half after the eagerly driven inner loop."""
    if not ENABLED or not torch.cuda.is_available():
        return None
    if not all(p.is_cuda for p in model.parameters()):
        return None
    fn = getattr(loss_fn, "__func__", None)
    cache = model.__dict__.setdefault("_mft_graph_steps", {})
    for name in ("set_forward_loss", "set_forward_loss_lockstep"):         # (lockstep: k episodes per step; the input shape is part of the key)
        if fn is not None and fn is getattr(type(model), name, None):
            st = cache.get(name)
            if st is None:
                st = cache[name] = GraphedLossBackward(model, loss_fn)
            return st
    if (fn is not None and fn is getattr(type(model), "set_forward_loss_finetune", None)
            and hasattr(model, "_finetune_prepare") and hasattr(model, "set_forward_loss_finetune_prepared")):
        st = cache.get("set_forward_loss_finetune")
        if st is None:
            st = cache["set_forward_loss_finetune"] = _PreparedStep(model)
        return st
    return None
