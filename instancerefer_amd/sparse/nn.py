"""torchsparse.nn surface used by the reference (models/basic_blocks.py:14-21,32-52;
models/attribute_module.py:20): Conv3d, BatchNorm, ReLU, GlobalMaxPooling — over libirx.so.

`conv_bn_act` is the fused entry the drop-in blocks call: conv -> BN statistics -> one apply pass
that also folds the residual add and the ReLU (3 kernels instead of torchsparse's 27x3 + 3)."""
import math

import torch
import torch.nn as nn

from . import functional as F_
from .tensor import SparseTensor


class Conv3d(nn.Module):
    """spnn.Conv3d: `kernel` (K, Cin, Cout) (or (Cin, Cout) for kernel_size 1), no bias by default,
    init U(+-1/sqrt(Cin*K)) — state-dict compatible with torchsparse checkpoints."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False,
                 transpose=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.dilation = dilation
        self.t = transpose
        if transpose:
            raise NotImplementedError("irx Conv3d: transposed convolution is not on the InstanceRefer path")
        if dilation != 1:
            raise NotImplementedError("irx Conv3d: dilation != 1 is not on the InstanceRefer path")
        if (kernel_size, stride) not in ((3, 1), (2, 2), (1, 1)):
            raise NotImplementedError("irx Conv3d: (kernel_size, stride)=(%s, %s) unsupported; the encoder uses "
                                      "(3,1) and (2,2)" % (kernel_size, stride))
        K = kernel_size ** 3
        self.kernel = nn.Parameter(torch.zeros(K, in_channels, out_channels) if kernel_size > 1
                                   else torch.zeros(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.init_weight()

    def __repr__(self):
        return "Conv3d(%d, %d, kernel_size=%d, stride=%d)" % (self.in_channels, self.out_channels,
                                                              self.kernel_size, self.stride)

    def init_weight(self):
        std = 1. / math.sqrt(self.in_channels * (self.kernel_size ** 3))
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    def conv_feats(self, x: SparseTensor):
        """-> (output features, output level)"""
        x = x.canonical()
        lv = x.level()
        if self.kernel_size == 1:
            y = x.F.matmul(self.kernel)
            out = lv
        elif self.kernel_size == 3:
            tbl, ld = lv.nbr27()
            y = F_.SparseConvFn.apply(x.F, self.kernel, tbl, ld, lv.n, lv.nbr27, 1, lv.pairs27)
            out = lv
        else:
            dm = lv.down()
            out = dm.out_level
            y = F_.SparseConvFn.apply(x.F, self.kernel, dm.child, dm.ld, out.n, dm.child_t, 0, dm.pairs)
        if self.bias is not None:
            y = y + self.bias
        return y, out

    def forward(self, x: SparseTensor):
        y, out = self.conv_feats(x)
        return SparseTensor(y, out.coords, out.stride, out.batch_size, out)


class BatchNorm(nn.BatchNorm1d):
    """spnn.BatchNorm = nn.BatchNorm1d over the rows of F."""

    def feats(self, f, residual=None, relu=False):
        if self.training or not self.track_running_stats:
            if self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            mom = 0.0 if self.momentum is None else self.momentum
            return F_.BatchNormActFn.apply(f, self.weight, self.bias, residual,
                                           self.running_mean if self.track_running_stats else None,
                                           self.running_var if self.track_running_stats else None,
                                           self.eps, mom, relu,
                                           F_.sync_group() if getattr(self, "_irx_sync", False) else None)
        return F_.bn_eval(f, self.weight, self.bias, residual, self.running_mean, self.running_var,
                          self.eps, relu)

    def forward(self, x):
        if isinstance(x, SparseTensor):
            return SparseTensor(self.feats(x.F), x.C, x.s, x._batch_size, x._level)
        return super().forward(x)


class ReLU(nn.ReLU):
    def forward(self, x):
        if isinstance(x, SparseTensor):
            return SparseTensor(torch.relu(x.F), x.C, x.s, x._batch_size, x._level)
        return super().forward(x)


class GlobalMaxPooling(nn.Module):
    """Per-batch-item channel-wise max of F -> (batch_size, C)."""

    def forward(self, x: SparseTensor):
        x = x.canonical()
        lv = x.level()
        return F_.segment_max(x.F, lv.offsets(), lv.batch_size)


def conv_bn_act(conv: Conv3d, bn: BatchNorm, x: SparseTensor, relu=True, residual: SparseTensor = None):
    """Fused Conv3d -> BatchNorm (-> + residual) (-> ReLU) on a SparseTensor."""
    y, out = conv.conv_feats(x)
    res = None
    if residual is not None:
        if residual._level is not out:
            raise RuntimeError("residual lives on a different coordinate level")
        res = residual.F
    y = bn.feats(y, res, relu)
    return SparseTensor(y, out.coords, out.stride, out.batch_size, out)
