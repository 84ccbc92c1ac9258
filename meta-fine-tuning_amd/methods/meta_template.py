"""MetaTemplate: the plugin base class of every few-shot method (mirror of methods/meta_template.py:10-186).

Host orchestration only -- episode loops, accuracy bookkeeping, the printed lines -- kept signature- and
behaviour-compatible with the reference so train.py / finetune.py style drivers run unchanged.  The
linear-head adaptation (set_forward_adaptation) runs its 100x7 SGD steps as HIP launches.
"""
from abc import abstractmethod

import numpy as np
import torch
import torch.nn as nn

from .. import graph_step, ops


class MetaTemplate(nn.Module):
    def __init__(self, model_func, n_way, n_support, change_way=True):
        super().__init__()
        self.n_way = n_way
        self.n_support = n_support
        self.n_query = -1                       # set per episode from the input
        self.freeze_backbone = False
        self.feature = model_func()
        self.feat_dim = self.feature.final_feat_dim
        self.change_way = change_way

    @abstractmethod
    def set_forward(self, x, is_feature):
        pass

    @abstractmethod
    def set_forward_loss(self, x):
        pass

    def forward(self, x):
        return self.feature.forward(x)

    def parse_feature(self, x, is_feature):
        x = x.cuda()
        if is_feature:
            z_all = x
        else:
            if self.freeze_backbone:
                for p in self.feature.parameters():
                    p.requires_grad = False
            x = x.contiguous().view(self.n_way * (self.n_support + self.n_query), *x.size()[2:])
            z_all = self.feature.forward(x).view(self.n_way, self.n_support + self.n_query, -1)
        return z_all[:, :self.n_support], z_all[:, self.n_support:]

    def correct(self, x):
        scores = self.set_forward(x)
        y_query = np.repeat(range(self.n_way), self.n_query)
        pred = scores.data.topk(1, 1, True, True)[1].cpu().numpy()
        return float(np.sum(pred[:, 0] == y_query)), len(y_query)

    # ------------------------------------------------------------------ episode loops
    def _episode_loop(self, epoch, train_loader, optimizer, loss_fn, support_from_x=True, n_support_images=None, lockstep=False):
        """``n_support_images``: support images per class in x when that is not ``self.n_support`` (gnnnet_copy's literal 50).
        ``lockstep``: x carries k episodes [k, n_way, n_support + n_query, ...] (train_loop_lockstep)."""
        print_freq = 10
        avg_loss = 0
        graphed = graph_step.for_loop(self, loss_fn)      # forward + backward as one hipGraph replay (plain set_forward_loss only)
        avg_dev = None
        # the running sum the reference keeps in a Python float (= double): the loss module adds every step's loss to a float64
        # device scalar inside its own launch (AG.CrossEntropyLoss.loss_sum) -- no extra launch and no host sync per step
        acc = getattr(getattr(self, "loss_fn", None), "loss_sum", None)
        acc = acc(next(self.parameters()).device) if (acc is not None and graphed is not None) else None
        base = acc.item() if acc is not None else 0.0
        for i, (x, _) in enumerate(train_loader):
            ep = 1 if lockstep else 0                    # (dimension of x that counts the classes)
            self.n_query = x.size(ep + 1) - (self.n_support if n_support_images is None else n_support_images)
            if self.change_way:
                self.n_way = x.size(ep)
            if graphed is not None:
                loss = graphed(x, optimizer)     # (grads of parameters outside the recorded step are dropped as zero_grad() would)
                optimizer.step()
                if acc is None:                  # (a loss module that is not ours: keep the sum on the device the slow way)
                    avg_dev = loss.detach().double() if avg_dev is None else avg_dev + loss.detach().double()
                if i % print_freq == 0:
                    tot = (acc.item() - base) if acc is not None else avg_dev.item()
                    print('Epoch {:d} | Batch {:d}/{:d} | Loss {:f}'.format(epoch, i, len(train_loader), tot / float(i + 1)))
                continue
            optimizer.zero_grad()
            loss = loss_fn(x)
            loss.backward()
            optimizer.step()
            avg_loss = avg_loss + loss.item()
            if i % print_freq == 0:
                print('Epoch {:d} | Batch {:d}/{:d} | Loss {:f}'.format(epoch, i, len(train_loader), avg_loss / float(i + 1)))

    def train_loop(self, epoch, train_loader, optimizer):
        self._episode_loop(epoch, train_loader, optimizer, self.set_forward_loss)

    def train_loop2(self, epoch, train_loader, optimizer):
        self._episode_loop(epoch, train_loader, optimizer, self.set_forward_loss)

    def train_loop_lockstep(self, epoch, train_loader, optimizer, k):
        """Opt-in (train.py --episodes_per_rank k; not in the reference): ONE optimizer step per k consecutive episodes of the
        loader, on the mean of their losses -- the update a k-rank episode-parallel run makes after its gradient all-reduce
        (SURVEY.md section 8(e)), with the k episodes running in lockstep through one sequence of launches (per-episode BatchNorm
        statistics).  The printed running loss is the mean over steps of the k-episode mean."""
        self._episode_loop(epoch, LockstepLoader(train_loader, k), optimizer, self.set_forward_loss_lockstep, lockstep=True)

    def train_loop_finetune(self, epoch, train_loader, optimizer):
        self._episode_loop(epoch, train_loader, optimizer, self.set_forward_loss_finetune)

    def train_loop3(self, epoch, train_loader, optimizer, unsup_loader):
        self._episode_loop(epoch, train_loader, optimizer, self.set_forward_loss)

    def test_loop(self, test_loader, record=None):
        acc_all = []
        iter_num = len(test_loader)
        for i, (x, _) in enumerate(test_loader):
            self.n_query = x.size(1) - self.n_support
            if self.change_way:
                self.n_way = x.size(0)
            correct_this, count_this = self.correct(x)
            acc_all.append(correct_this / count_this * 100)
        acc_all = np.asarray(acc_all)
        acc_mean = np.mean(acc_all)
        acc_std = np.std(acc_all)
        print('%d Test Acc = %4.2f%% +- %4.2f%%' % (iter_num, acc_mean, 1.96 * acc_std / np.sqrt(iter_num)))
        return acc_mean

    # ------------------------------------------------------------------ linear-head adaptation
    def set_forward_adaptation(self, x, is_feature=True):
        """Fix the features, train a fresh Linear(feat_dim, n_way) with SGD(lr .01, momentum .9, dampening .9,
        wd 1e-3) for 100 epochs of 4-sample batches (meta_template.py:153-186; baselinefinetune.py:17-58)."""
        assert is_feature == True, 'Feature is fixed in further adaptation'  # noqa: E712
        z_support, z_query = self.parse_feature(x, is_feature)
        z_support = z_support.contiguous().view(self.n_way * self.n_support, -1).float()
        z_query = z_query.contiguous().view(self.n_way * self.n_query, -1).float()
        y_support = np.repeat(range(self.n_way), self.n_support).astype(np.int32)
        linear_clf = nn.Linear(self.feat_dim, self.n_way).cuda()       # same torch-RNG draw as the reference
        return linear_head_adapt(z_support, y_support, z_query, linear_clf.weight.data, linear_clf.bias.data,
                                 self.n_way, self.n_support)


class LockstepLoader:
    """k consecutive episodes of an episode loader as one batch [k, n_way, n_support + n_query, 3, H, W] (a trailing group of fewer
    than k episodes is not drawn: every step has the same shape, and every rank of a multi-GPU run the same number of steps)."""

    def __init__(self, loader, k):
        self.loader, self.k = loader, int(k)
        assert self.k >= 1

    def __len__(self):
        return len(self.loader) // self.k

    def __iter__(self):
        buf = []
        for x, y in self.loader:
            buf.append(x)
            if len(buf) == self.k:
                yield torch.stack(buf), None
                buf = []


def linear_head_adapt(z_support, y_support, z_query, w, b, n_way, n_support, epochs=100, batch_size=4):
    """SGD-with-dampening head training (meta_template.py:160-186) as ONE launch: the permutations of all epochs are drawn
    first (same numpy stream order as the reference's per-epoch draws), uploaded as an index table, and
    mft_linear_head_sgd_run executes the 100 x 7 dependent steps inside a single workgroup; then one GEMM scores the queries."""
    dev = z_support.device
    K = z_support.shape[1]
    assert K % 32 == 0
    support_size = n_way * n_support
    steps = []
    for epoch in range(epochs):
        rand_id = np.random.permutation(support_size)
        for i in range(0, support_size, batch_size):
            ids = rand_id[i:min(i + batch_size, support_size)]
            steps.append(np.concatenate([ids, -np.ones(batch_size - len(ids), dtype=ids.dtype)]))
    table = torch.from_numpy(np.stack(steps).astype(np.int32)).to(dev)
    W = w.detach().clone().float().contiguous().view(1, n_way, K)
    bias = b.detach().clone().float().contiguous().view(1, n_way)
    y_dev = torch.from_numpy(np.asarray(y_support).astype(np.int32)).to(dev)
    rc = ops._lib.lib().mft_linear_head_sgd_run(ops._p(z_support.contiguous()), ops._p(y_dev), ops._p(table), 1, support_size, K,
                                                n_way, table.shape[0], batch_size, ops._p(W), ops._p(bias), 0.01, 0.9, 0.9, 0.001,
                                                ops._stream())
    ops._lib.check(rc, "mft_linear_head_sgd_run")
    wpad = torch.zeros((32, K), device=dev)               # rows >= n_way stay zero (N padded to the 32-wide tile)
    wpad[:n_way] = W[0]
    return ops.gemm(z_query, K, wpad[:n_way].contiguous(), n_way, bias=bias[0].contiguous())


def small_tn(a, b):
    """a [k, m], b [k, n] -> a^T b [m, n] through the MFMA GEMM (k padded to 32 with zero rows)."""
    k, m = a.shape
    n = b.shape[1]
    at = torch.zeros((m, 32), device=a.device)
    at[:, :k] = a.t()
    bt = torch.zeros((n, 32), device=a.device)
    bt[:, :k] = b.t()
    return ops.gemm(at.contiguous(), 32, bt.contiguous(), n)
