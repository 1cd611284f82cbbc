"""SymmetricConv2d on MI355X — same constructor, parameters and state_dict as the reference
layer (reference symmetric_layers_torch.py:21-138), forward/backward on the HIP engine.

The reference rebuilds the full filter bank every forward with torch.flip + torch.cat and
hands it to ATen's conv2d; here the mirrored copies are never materialised in the reference
layout: the bank packer reads filter j >= U as its unique source with kx and / or ky reversed
('h' pairs, 'v' pairs, 'hv' quadruples in the reference's torch.cat order), and the filter
gradient folds every mirrored copy's dW (flipped back) onto the unique filter's.
"""
import math

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import _lib as L
from .engine import single_layer_graph
from .hipnet import HipNetMixin


def _same_padding(kernel_size, padding, dilation):
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    if isinstance(padding, str):
        if padding == "same":
            return (k - 1) // 2 * (dilation if isinstance(dilation, int) else dilation[0])
        if padding == "valid":
            return 0
        raise ValueError(padding)
    return padding if isinstance(padding, int) else padding[0]


class SymmetricConv2d(nn.Conv2d, HipNetMixin):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = "zeros", symmetry: dict = {},
                 share_bias: bool = False):
        """symmetry: {'h': n, 'v': n, 'hv': n} — numbers of filters that are mirror images of other
        filters about the horizontal axis / vertical axis / both (reference :36-44)."""
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         padding_mode)
        if self.groups > 1:
            raise ValueError(self.__str__() + " does not support groups>1")
        self.share_bias = share_bias if bias else False   # stored, never applied (as in the reference)
        if symmetry is None:
            self.symmetry = None
            self.unique_out_channels = out_channels
        else:
            symmetry = dict(symmetry)
            for key in ("h", "v", "hv"):
                symmetry.setdefault(key, 0)
            self.symmetry = symmetry
            for key, val in symmetry.items():
                if key in ("h", "v") and val % 2 != 0:
                    raise ValueError("Number of symmetric h and v filters must be divisible by 2")
                if key == "hv" and val % 4 != 0:
                    raise ValueError("Number of symmetric hv filters must be divisible by 4")
            assert sum(symmetry.values()) <= self.out_channels, \
                "Number of symmetric channels exceeds number of out channels"
            self.unique_out_channels = (self.out_channels - symmetry["h"] // 2 - symmetry["v"] // 2
                                        - 3 * symmetry["hv"] // 4)
            self.weight = Parameter(torch.empty(self.unique_out_channels, in_channels, *self.kernel_size))
            if bias:
                self.bias = Parameter(torch.empty(out_channels))
        self.reset_parameters()
        self._graph_built = False

    def _build_graph(self):
        k = self.kernel_size[0]
        if self.kernel_size[0] != self.kernel_size[1] or k not in (3, 5):
            raise NotImplementedError("HIP SymmetricConv2d supports square 3x3 / 5x5 kernels")
        if tuple(self.stride) != (1, 1) or tuple(self.dilation) != (1, 1):
            raise NotImplementedError("HIP SymmetricConv2d supports stride 1, dilation 1")
        if self.bias is None:
            raise NotImplementedError("HIP SymmetricConv2d expects bias=True (as every reference call site)")
        s = self.symmetry or {"h": 0, "v": 0, "hv": 0}
        pad = _same_padding(self.kernel_size, self.padding, self.dilation)
        self._init_hipnet(single_layer_graph(self.in_channels, self.out_channels, k, pad, self.padding_mode, s["h"],
                                             L.POST_NONE, "none", 1, gn=False, sym_v=s["v"], sym_hv=s["hv"]))
        self._graph_built = True

    def forward(self, input):
        if not self._graph_built:
            self._build_graph()
        return self._run_graph(input)
