"""torchsparse.nn modules restated: Conv3d (kernel (K,Cin,Cout), no bias, U(+-1/sqrt(Cin*K)) init),
BatchNorm (= nn.BatchNorm1d over F), ReLU, GlobalMaxPooling."""
import math

import torch
import torch.nn as nn

from ..tensor import SparseTensor
from . import emulate
from . import functional as spf


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False,
                 transpose=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.dilation = dilation
        self.kernel = nn.Parameter(torch.zeros(kernel_size ** 3, in_channels, out_channels)) \
            if kernel_size > 1 else nn.Parameter(torch.zeros(in_channels, out_channels))
        self.bias = None if not bias else nn.Parameter(torch.zeros(out_channels))
        self.t = transpose
        self.init_weight()

    def init_weight(self):
        std = 1. / math.sqrt(self.out_channels if self.t else self.in_channels * (self.kernel_size ** 3))
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    def forward(self, inputs):
        return spf.conv3d(inputs, self.kernel, self.kernel_size, self.bias, self.stride, self.dilation, self.t)


class BatchNorm(nn.BatchNorm1d):
    def forward(self, inputs):
        out = super().forward(inputs.F)
        t = SparseTensor(out, inputs.C, inputs.s)
        t.coord_maps = inputs.coord_maps
        t.kernel_maps = inputs.kernel_maps
        return t


class ReLU(nn.ReLU):
    def forward(self, inputs):
        out = emulate.on_relu(super().forward(inputs.F))
        t = SparseTensor(out, inputs.C, inputs.s)
        t.coord_maps = inputs.coord_maps
        t.kernel_maps = inputs.kernel_maps
        return t


class GlobalMaxPooling(nn.Module):
    def forward(self, inputs):
        return spf.global_max_pool(inputs)
