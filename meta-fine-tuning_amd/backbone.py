"""ResNet10 backbone with the reference's module tree / state_dict contract and a HIP forward+backward.

Mirror of backbone.{init_layer, Flatten, SimpleBlock, ResNet, ResNet10} (backbone.py:9-23,216-261,
401-439,519-520).  The nn.Conv2d / nn.BatchNorm2d children exist as *parameter containers* so that
key names, ``named_parameters()`` order (the last 9 names are the inner-loop-adaptable set,
finetune.py:236-252), ``load_state_dict``, ``copy.deepcopy`` and optimisers behave exactly as in the
reference; their own ``forward`` is never used -- ``ResNet.forward`` runs the gfx950 kernels of
libmft_hip.so on NHWC activations.  There is no CPU path: a CPU input raises.
"""
import math

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from . import autograd_ops as AG


def init_layer(L):
    """Fan-out initialisation (backbone.py:9-16)."""
    if isinstance(L, nn.Conv2d):
        n = L.kernel_size[0] * L.kernel_size[1] * L.out_channels
        L.weight.data.normal_(0, math.sqrt(2.0 / float(n)))
    elif isinstance(L, nn.BatchNorm2d):
        L.weight.data.fill_(1)
        L.bias.data.fill_(0)


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class SimpleBlock(nn.Module):
    """Residual block container (backbone.py:216-261); children registered in the reference's order."""
    maml = False

    def __init__(self, indim, outdim, half_res):
        super().__init__()
        self.indim, self.outdim, self.half_res = indim, outdim, half_res
        self.C1 = nn.Conv2d(indim, outdim, kernel_size=3, stride=2 if half_res else 1, padding=1, bias=False)
        self.BN1 = nn.BatchNorm2d(outdim)
        self.C2 = nn.Conv2d(outdim, outdim, kernel_size=3, padding=1, bias=False)
        self.BN2 = nn.BatchNorm2d(outdim)
        self.relu1 = nn.ReLU(inplace=True)
        self.relu2 = nn.ReLU(inplace=True)
        self.parametrized_layers = [self.C1, self.C2, self.BN1, self.BN2]
        if indim != outdim:
            self.shortcut = nn.Conv2d(indim, outdim, 1, 2 if half_res else 1, bias=False)
            self.BNshortcut = nn.BatchNorm2d(outdim)
            self.parametrized_layers += [self.shortcut, self.BNshortcut]
            self.shortcut_type = '1x1'
        else:
            self.shortcut_type = 'identity'
        for layer in self.parametrized_layers:
            init_layer(layer)

    def forward(self, x):
        raise RuntimeError("SimpleBlock is executed by ResNet.forward on the HIP path, not on its own")


class ResNet(nn.Module):
    """ResNet container (backbone.py:401-439).  ``trunk`` keeps the reference's Sequential indices
    0 Conv2d, 1 BatchNorm2d, 2 ReLU, 3 MaxPool2d, 4-7 SimpleBlock, 8 AvgPool2d, 9 Flatten."""
    maml = False

    def __init__(self, block, list_of_num_layers, list_of_out_dims, flatten=False):
        super().__init__()
        assert len(list_of_num_layers) == 4, 'Can have only four stages'
        if list(list_of_num_layers) != [1, 1, 1, 1] or list(list_of_out_dims) != [64, 128, 256, 512]:
            raise NotImplementedError("only the ResNet10 geometry is built on the HIP path (SURVEY.md §2.1)")
        conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        bn1 = nn.BatchNorm2d(64)
        init_layer(conv1)
        init_layer(bn1)
        trunk = [conv1, bn1, nn.ReLU(), nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]
        indim = 64
        for i in range(4):
            for j in range(list_of_num_layers[i]):
                half_res = (i >= 1) and (j == 0)
                trunk.append(block(indim, list_of_out_dims[i], half_res))
                indim = list_of_out_dims[i]
        if flatten:
            # the reference's nn.AvgPool2d(7) is a *global* pool at 224x224; executed as a global average
            # pool so that 84x84 inputs (3x3 map) work too
            trunk.append(nn.AvgPool2d(7))
            trunk.append(Flatten())
            self.final_feat_dim = indim
        else:
            self.final_feat_dim = [indim, 7, 7]
        self.flatten = flatten
        self.trunk = nn.Sequential(*trunk)

    def forward(self, x):
        if not self.flatten:
            raise NotImplementedError("flatten=False feature maps are off the GNN hot path")
        return AG.resnet10_module_forward(self, x)


def ResNet10(flatten=True):
    return ResNet(SimpleBlock, [1, 1, 1, 1], [64, 128, 256, 512], flatten)
