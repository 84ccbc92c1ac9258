"""GNN few-shot head: gmul / Gconv / Wcompute / GNN_nl with the reference's module tree and state_dict keys
(mirror of methods/gnn.py:16-166), executed by the HIP kernels (functional.gnn_forward).

The nn.Conv2d / nn.BatchNorm / nn.Linear children are parameter containers; forward passes go to
libmft_hip.so.  BatchNorm layers use batch statistics in train and eval alike
(track_running_stats=False, gnn.py:40,65-74).
"""
import torch
import torch.nn as nn

from .. import autograd_ops as AG
from .. import functional as Fn
from .. import ops


def gmul(input):
    """(W [bs,N,N,J], x [bs,N,F]) -> [bs,N,J*F] = cat_j(W[..., j] @ x)  (gnn.py:16-28)."""
    W, x = input
    AG._require_cuda(x, "gmul")
    bs, N, F = x.shape
    outs = []
    xp = x.contiguous().view(bs * N, F).float()
    for j in range(W.shape[3]):
        y = ops.graph_aggregate(W[..., j].contiguous().float(), xp, F, ops.round_up(2 * F, 4))
        outs.append(y[:, F:2 * F].reshape(bs, N, F))
    return torch.cat(outs, 2)


class Gconv(nn.Module):
    maml = False

    def __init__(self, nf_input, nf_output, J, bn_bool=True):
        super().__init__()
        if self.maml:
            raise NotImplementedError("fast-weight (gnnnet_maml) layers are off the hot path (SURVEY.md §2.1)")
        self.J = J
        self.num_inputs = J * nf_input
        self.num_outputs = nf_output
        self.fc = nn.Linear(self.num_inputs, self.num_outputs)
        self.bn_bool = bn_bool
        if self.bn_bool:
            self.bn = nn.BatchNorm1d(self.num_outputs, track_running_stats=False)

    def forward(self, input):
        W, x = input
        AG._require_cuda(x, "Gconv")
        bs, N, F = x.shape
        G = AG.solo_weights(self, "gc")
        arena = AG.arena_for(x.device)
        xp = AG.pad_rows(x.contiguous().view(bs * N, F).float(), 256)
        o = Fn.gconv(G, "solo", W[..., 1].contiguous().float(), xp, F, bs, N, 1, arena, tag="gconv_mod")
        return W, o.clone().view(bs, N, self.num_outputs)


class Wcompute(nn.Module):
    maml = False

    def __init__(self, input_features, nf, operator='J2', activation='softmax', ratio=[2, 2, 1, 1], num_operators=1,
                 drop=False):
        super().__init__()
        if self.maml:
            raise NotImplementedError("fast-weight (gnnnet_maml) layers are off the hot path (SURVEY.md §2.1)")
        if operator != 'J2' or activation != 'softmax' or drop:
            raise NotImplementedError("only operator='J2', activation='softmax', drop=False is used by GNN_nl")
        self.num_features = nf
        self.operator = operator
        self.conv2d_1 = nn.Conv2d(input_features, int(nf * ratio[0]), 1, stride=1)
        self.bn_1 = nn.BatchNorm2d(int(nf * ratio[0]), track_running_stats=False)
        self.drop = drop
        self.conv2d_2 = nn.Conv2d(int(nf * ratio[0]), int(nf * ratio[1]), 1, stride=1)
        self.bn_2 = nn.BatchNorm2d(int(nf * ratio[1]), track_running_stats=False)
        self.conv2d_3 = nn.Conv2d(int(nf * ratio[1]), nf * ratio[2], 1, stride=1)
        self.bn_3 = nn.BatchNorm2d(nf * ratio[2], track_running_stats=False)
        self.conv2d_4 = nn.Conv2d(nf * ratio[2], nf * ratio[3], 1, stride=1)
        self.bn_4 = nn.BatchNorm2d(nf * ratio[3], track_running_stats=False)
        self.conv2d_last = nn.Conv2d(nf, num_operators, 1, stride=1)
        self.activation = activation

    def forward(self, x, W_id):
        AG._require_cuda(x, "Wcompute")
        bs, N, F = x.shape
        G = AG.solo_weights(self, "wc")
        xp = AG.pad_rows(x.contiguous().view(bs * N, F).float(), 256)
        A = Fn.wcompute(G, "solo", xp, F, bs, N, 1, AG.arena_for(x.device), tag="wc_mod")
        return torch.cat([W_id, A.clone().unsqueeze(3)], 3)


class GNN_nl(nn.Module):
    def __init__(self, input_features, nf, train_N_way):
        super().__init__()
        self.input_features = input_features
        self.nf = nf
        self.num_layers = 2
        for i in range(self.num_layers):
            fin = self.input_features + int(nf / 2) * i
            self.add_module('layer_w{}'.format(i), Wcompute(fin, nf, operator='J2', activation='softmax', ratio=[2, 2, 1, 1]))
            self.add_module('layer_l{}'.format(i), Gconv(fin, int(nf / 2), 2))
        fin = self.input_features + int(self.nf / 2) * self.num_layers
        self.w_comp_last = Wcompute(fin, nf, operator='J2', activation='softmax', ratio=[2, 2, 1, 1])
        self.layer_last = Gconv(fin, train_N_way, 2, bn_bool=False)
        self.train_N_way = train_N_way

    def forward(self, x):
        """x [n_graphs, N, 128+n_way] -> [n_graphs, N, n_way] (gnn.py:154-166)."""
        AG._require_cuda(x, "GNN_nl")
        bs, N, F = x.shape
        G = AG.head_weights(self)
        nodes = AG.pad_rows(x.contiguous().view(bs * N, F).float(), 256)
        out = Fn.gnn_forward(G, nodes, bs, N, 1, AG.arena_for(x.device), tag="gnn_mod")
        return out.clone().view(bs, N, -1)
