"""GnnNet: ResNet10 features -> Linear+BN projector -> GNN over (support + one query) graphs
(mirror of methods/gnnnet.py:20-231), with the reference's attributes, asserts and state_dict keys.

All arithmetic runs in libmft_hip.so (the loss included: autograd_ops.CrossEntropyLoss).  ``set_forward`` batches the n_query graphs of an episode in one
grouped launch sequence; ``set_forward_finetune`` runs the first-order-MAML inner loop (15 epochs of
4-image Adam steps on the last ResNet block, gnnnet.py:153-177) on device-resident per-episode state.
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from .. import autograd_ops as AG
from .. import backbone
from .. import engine as eng
from .. import functional as Fn
from .. import ops
from .gnn import GNN_nl
from .meta_template import MetaTemplate


class Classifier(nn.Module):
    """Constructed (and never trained) by the reference's inner loop (gnnnet.py:10-18,127); kept so that the
    torch RNG stream matches when callers depend on it."""

    def __init__(self, dim, n_way):
        super().__init__()
        self.fc = nn.Linear(dim, n_way)

    def forward(self, x):
        raise RuntimeError("Classifier receives no gradient on the hot path (SURVEY.md §0 D4)")


class GnnNet(MetaTemplate):
    maml = False
    FOLD50 = False          # gnnnet_copy.GnnNet folds 50 supports to 25 graph nodes per class

    def __init__(self, model_func, n_way, n_support):
        super().__init__(model_func, n_way, n_support)
        if self.maml:
            raise NotImplementedError("gnnnet_maml fast-weight layers are off the hot path (SURVEY.md §2.1)")
        self.loss_fn = AG.CrossEntropyLoss()                         # nn.CrossEntropyLoss() (gnnnet.py:43) on mft_cross_entropy_mean
        self.first = True
        self.fc = nn.Sequential(nn.Linear(self.feat_dim, 128), nn.BatchNorm1d(128, track_running_stats=False))
        self.gnn = GNN_nl(128 + self.n_way, 96, self.n_way)
        self.method = 'GnnNet'
        self.support_label = _support_label(self.n_way, self._graph_support())

    def _graph_support(self):
        """support nodes per class in the graph"""
        return self.n_support

    def _image_support(self):
        """support images per class in an episode tensor"""
        return self.n_support

    def cuda(self):
        self.feature.cuda()
        self.fc.cuda()
        self.gnn.cuda()
        self.support_label = self.support_label.cuda()
        return self

    # ------------------------------------------------------------------ forward
    def set_forward(self, x, is_feature=False):
        """x [n_way, n_support+n_query, 3,H,W] (or features [n_way, n_support+15, 512]) -> scores
        [n_way*n_query, n_way], row = class*n_query + q (gnnnet.py:68-87)."""
        x = x.cuda()
        if is_feature:
            assert (x.size(1) == self.n_support + 15)
            feats = x.reshape(-1, x.size(-1))
        else:
            feats = self.feature(x.view(-1, *x.size()[2:]))
        n_query = x.size(1) - self.n_support if is_feature else self.n_query
        return AG.gnnnet_head(self, feats, self._graph_support(), n_query, fold=self.FOLD50)

    def forward_gnn(self, zs):
        """zs: list of n_query tensors [1, n_way*(n_support+1), 128] -> scores (gnnnet.py:210-217)."""
        nodes = torch.cat([torch.cat([z, self.support_label], dim=2) for z in zs], dim=0)
        scores = self.gnn(nodes)
        ns = self._graph_support()
        return scores.view(self.n_query, self.n_way, ns + 1, self.n_way)[:, :, -1].permute(1, 0, 2).contiguous().view(-1, self.n_way)

    def _y_query(self):
        """np.repeat(range(n_way), n_query) on the device (gnnnet.py:220), uploaded once per (n_way, n_query): the per-step upload
        is a synchronous copy, which a hipGraph capture of the step refuses."""
        key = (self.n_way, self.n_query, torch.cuda.current_device())
        y = self._yq_cache.get(key) if hasattr(self, "_yq_cache") else None
        if y is None:
            if not hasattr(self, "_yq_cache"):
                self._yq_cache = {}
            y = self._yq_cache[key] = torch.from_numpy(np.repeat(range(self.n_way), self.n_query)).cuda()
        return y

    def set_forward_loss(self, x):
        y_query = self._y_query()
        scores = self.set_forward(x)
        return self.loss_fn(scores, y_query)

    # ------------------------------------------------------------------ k episodes in lockstep (opt-in, train.py --episodes_per_rank k)
    def set_forward_lockstep(self, xs):
        """xs [k, n_way, n_support+n_query, 3,H,W]: k episodes through ONE sequence of launches -- every BatchNorm of the backbone
        and of the head keeps per-episode statistics (a group per episode), convolutions / linear layers see k times the rows.
        Scores [k*n_way*n_query, n_way], episode after episode, each exactly what ``set_forward`` gives for that episode."""
        xs = xs.cuda()
        k = xs.size(0)
        feats = AG.resnet10_module_forward(self.feature, xs.reshape(-1, *xs.size()[3:]), groups=k)
        return AG.gnnnet_head(self, feats, self._graph_support(), self.n_query, fold=self.FOLD50, episodes=k)

    def set_forward_loss_lockstep(self, xs):
        """mean over the k episodes of ``set_forward_loss`` (gnnnet.py:219-224): its backward leaves the AVERAGE of the k episodes'
        gradients taken at the same parameters -- the step a k-rank episode-parallel run takes after its all-reduce
        (parallel.FlatGradBucket), here on one GPU with k times the work per launch."""
        k = xs.size(0)
        key = ("lockstep", self.n_way, self.n_query, k, torch.cuda.current_device())
        if not hasattr(self, "_yq_cache"):
            self._yq_cache = {}
        y = self._yq_cache.get(key)
        if y is None:
            y = self._yq_cache[key] = torch.from_numpy(np.tile(np.repeat(range(self.n_way), self.n_query), k)).cuda()
        return self.loss_fn(self.set_forward_lockstep(xs), y)

    def set_forward_loss_finetune(self, x):
        y_query = self._y_query()
        scores = self.set_forward_finetune(x)
        return self.loss_fn(scores, y_query)

    # ------------------------------------------------------------------ first-order MAML
    def MAML_update(self):
        """theta <- theta - (theta_adapted_prev - theta_pre_prev) on the last nine tensors (gnnnet.py:90-103)."""
        if self.first:
            return
        names = [n for n, p in self.feature.named_parameters() if p.requires_grad]
        keep = set(names[:-9])
        for (name, p), (_, p2), (_, p3) in zip(self.feature.named_parameters(), self.feature2.named_parameters(),
                                               self.feature3.named_parameters()):
            if name not in keep:
                ops.maml_delta(p.data, p2.data.contiguous(), p3.data.contiguous())
        AG.touch(self.feature)                 # p.data was written by a kernel: packed copies refreshed in place at the next use

    INNER_EPOCHS = 15

    def set_forward_finetune(self, x, is_feature=False):
        """Meta-fine-tuning forward (gnnnet.py:106-208): undo the previous episode's inner-loop delta, adapt a
        copy of the backbone's last block on the support set, swap it in, then score support (BN batch 25) and
        query (BN batch 80) separately through fc + GNN."""
        x = x.cuda()
        self._finetune_prepare(x)
        return self._finetune_scores(x)

    def set_forward_loss_finetune_prepared(self, x):
        """The differentiable half of set_forward_loss_finetune, after ``_finetune_prepare(x)``: what the episode loop replays from
        a hipGraph (graph_step.for_loop)."""
        return self.loss_fn(self._finetune_scores(x.cuda()), self._y_query())

    def _finetune_prepare(self, x):
        """gnnnet.py:106-187: MAML_update, the inner loop on the support set, theta_pre / theta_adapted, the backbone takes the
        adapted state.  Host-driven (numpy permutations); its launches are recorded separately (engine.adapt_last_block)."""
        batch_size = 4
        n_sup = self._image_support()
        support_size = self.n_way * n_sup
        for p in self.feature.parameters():
            p.requires_grad = True
        y_a = np.repeat(range(self.n_way), n_sup).astype(np.int32)
        self.MAML_update()
        x_a = x[:, :n_sup].contiguous().view(support_size, *x.size()[2:])
        Classifier(self.feat_dim, self.n_way)                      # RNG-stream parity only (gnnnet.py:127)
        adapted = eng.adapt_last_block(self.feature, x_a, y_a, epochs=self.INNER_EPOCHS, batch_size=batch_size)
        if self.first:
            self.first = False
        # theta_pre = copy of the backbone, theta_adapted = that copy with the inner loop's result loaded, and the backbone itself
        # takes theta_adapted's state (gnnnet.py:185-187).  Same values with two multi-tensor copies into the holders created on
        # the first episode instead of two deepcopies + two full load_state_dicts per episode (12 ms of 73).
        self.feature2 = _copy_module(getattr(self, "feature2", None), self.feature)     # theta_pre      (gnnnet.py:185)
        self.feature.load_state_dict(adapted, strict=False)        # nine tensors + BN running stats, in place (version counters bump)
        self.feature3 = _copy_module(getattr(self, "feature3", None), self.feature)     # theta_adapted  (gnnnet.py:186)
        for p in self.feature.parameters():
            p.requires_grad = True

    def _finetune_scores(self, x):
        """gnnnet.py:189-208: support (BN batch 25) and query (BN batch 80) through the adapted backbone, fc + GNN."""
        n_sup = self._image_support()
        support_size = self.n_way * n_sup
        x_b = x[:, n_sup:].contiguous().view(self.n_way * self.n_query, *x.size()[2:])
        x_a = x[:, :n_sup].contiguous().view(support_size, *x.size()[2:])
        out_s = self.feature(x_a).view(self.n_way, n_sup, -1)
        out_q = self.feature(x_b).view(self.n_way, self.n_query, -1)
        final = torch.cat((out_s, out_q), dim=1)
        assert (final.size(1) == n_sup + 16)
        return AG.gnnnet_head(self, final.view(-1, final.size(-1)), self._graph_support(), self.n_query, fold=self.FOLD50)


def _copy_module(dst, src):
    """``copy.deepcopy(src)`` the first time; afterwards the same values written into the existing holder with one multi-tensor
    copy for the parameters and one per buffer dtype."""
    if dst is None:
        return copy.deepcopy(src)
    dp, sp = list(dst.parameters()), list(src.parameters())
    db, sb = list(dst.buffers()), list(src.buffers())
    if len(dp) != len(sp) or len(db) != len(sb) or any(a.shape != b.shape for a, b in zip(dp + db, sp + sb)):
        return copy.deepcopy(src)
    with torch.no_grad():
        torch._foreach_copy_(dp, sp)
        for dt in {a.dtype for a in db}:
            torch._foreach_copy_([a for a in db if a.dtype == dt], [b for a, b in zip(db, sb) if a.dtype == dt])
    return dst


def _support_label(n_way, n_support):
    """One-hot rows for supports, a zero row per query slot: [1, n_way*(n_support+1), n_way] (gnnnet.py:34-38)."""
    lab = torch.zeros(n_way, n_support + 1, n_way)
    for c in range(n_way):
        lab[c, :n_support, c] = 1.0
    return lab.view(1, -1, n_way)
