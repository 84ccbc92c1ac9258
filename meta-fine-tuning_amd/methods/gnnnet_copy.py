"""50-shot "compressed GNN" variant (mirror of methods/gnnnet_copy.py:21-271): the graph carries
round(50/2)=25 support nodes per class, each the mean of supports k and k+25; the inner loop runs 5 epochs
over all 50 supports per class."""
from .. import autograd_ops as AG
from .gnnnet import GnnNet as _GnnNet, _support_label


class GnnNet(_GnnNet):
    FOLD50 = True
    INNER_EPOCHS = 5                                   # gnnnet_copy.py:177

    def __init__(self, model_func, n_way, n_support):
        super().__init__(model_func, n_way, n_support)
        self.n_support = round(self.n_support / 2)     # gnnnet_copy.py:34 -- attribute holds the *graph* support count
        self.support_label = _support_label(self.n_way, self.n_support)

    def _image_support(self):
        return self.n_support * 2

    def set_forward(self, x, is_feature=False):
        x = x.cuda()
        true_ns = self.n_support * 2
        if is_feature:
            assert (x.size(1) == true_ns + 15)
            feats = x.reshape(-1, x.size(-1))
            n_query = x.size(1) - true_ns
        else:
            feats = self.feature(x.view(-1, *x.size()[2:]))
            n_query = self.n_query
        return AG.gnnnet_head(self, feats, self.n_support, n_query, fold=True)

    def _loop50(self, epoch, train_loader, optimizer, loss_fn):
        # n_query = x.size(1) - 50, the literal of gnnnet_copy.py:86,104; otherwise MetaTemplate's loop (incl. its hipGraph replay)
        self._episode_loop(epoch, train_loader, optimizer, loss_fn, n_support_images=50)

    def train_loop50(self, epoch, train_loader, optimizer):
        self._loop50(epoch, train_loader, optimizer, self.set_forward_loss)

    def train_loop_finetune50(self, epoch, train_loader, optimizer):
        self._loop50(epoch, train_loader, optimizer, self.set_forward_loss_finetune)
