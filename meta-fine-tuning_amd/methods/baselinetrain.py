"""Supervised pre-training of the "baseline" checkpoint of the README ensemble (mirror of methods/baselinetrain.py:10-59):
backbone features -> nn.Linear(512, num_class) -> cross entropy, ordinary mini-batches.  The backbone runs through the
HIP forward / full backward (autograd_ops), the classifier through the same GEMM / weight-gradient kernels
(functional_bwd._linear_fwd/_linear_bwd); `loss_type='dist'` (Baseline++, backbone.distLinear) is outside the hot path."""
import torch
import torch.nn as nn

from .. import autograd_ops as AG
from .. import functional_bwd as FB
from .. import ops


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b on the fp32-MFMA GEMM; backward = data gradient, weight gradient and column sum launches."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        cout, K = w.shape
        wpk = ops.pack_conv_weight(w.detach())
        o = FB._linear_fwd(x, K, wpk, b.detach().contiguous(), cout)
        ctx.save_for_backward(x, wpk)
        ctx.dims = (cout, K, o.shape[1])
        return o[:, :cout].clone()

    @staticmethod
    def backward(ctx, dy):
        x, wpk = ctx.saved_tensors
        cout, K, cp = ctx.dims
        d_o = torch.zeros((x.shape[0], cp), device=x.device, dtype=torch.float32)
        d_o[:, :cout] = dy
        dx, dW, db = FB._linear_bwd(x, K, wpk, d_o, cout, need_dx=ctx.needs_input_grad[0])
        return dx, dW[:, :K].contiguous(), db


class AverageMeter(object):
    """utils.AverageMeter (utils.py:12-27)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class BaselineTrain(nn.Module):
    def __init__(self, model_func, num_class, loss_type='softmax'):
        super(BaselineTrain, self).__init__()
        if loss_type != 'softmax':
            raise NotImplementedError("loss_type='dist' (Baseline++) is outside the HIP hot path")
        self.feature = model_func()
        self.classifier = nn.Linear(self.feature.final_feat_dim, num_class)
        self.classifier.bias.data.fill_(0)                               # baselinetrain.py:17
        self.loss_type = loss_type
        self.num_class = num_class
        self.loss_fn = AG.CrossEntropyLoss()                             # nn.CrossEntropyLoss() (baselinetrain.py:20), HIP launches
        self.top1 = AverageMeter()

    def forward(self, x):
        out = self.feature.forward(x.cuda())
        return _LinearFn.apply(out, self.classifier.weight, self.classifier.bias)

    def forward_loss(self, x, y):
        y = y.cuda()
        scores = self.forward(x)
        _, predicted = torch.max(scores.data, 1)
        correct = predicted.eq(y.data).cpu().sum()
        self.top1.update(correct.item() * 100 / (y.size(0) + 0.0), y.size(0))
        return self.loss_fn(scores, y)

    def train_loop(self, epoch, train_loader, optimizer):
        print_freq = 10
        avg_loss = 0
        for i, (x, y) in enumerate(train_loader):
            optimizer.zero_grad()
            loss = self.forward_loss(x, y)
            loss.backward()
            optimizer.step()
            avg_loss = avg_loss + loss.item()
            if i % print_freq == 0:
                print('Epoch {:d} | Batch {:d}/{:d} | Loss {:f} | Top1 Val {:f} | Top1 Avg {:f}'.format(
                    epoch, i, len(train_loader), avg_loss / float(i + 1), self.top1.val, self.top1.avg))

    def test_loop(self, val_loader):
        return -1                                                        # baselinetrain.py:58: no validation
