"""torchsparse.SparseTensor restated: feats F (N,C), coords C (N,4) = (x,y,z,batch), stride s,
cached coord_maps / kernel_maps that ride along through BN / ReLU / `+`."""
import numpy as np
import torch


class SparseTensor:
    def __init__(self, feats, coords, cur_tensor_stride=1):
        self.F = feats
        self.C = coords
        self.s = cur_tensor_stride
        self.coord_maps = {}
        self.kernel_maps = {}

    def check(self):
        if self.s not in self.coord_maps:
            self.coord_maps[self.s] = self.C

    def cuda(self):
        # oracle runs on CPU: `.cuda()` is the identity (reference hard-codes it)
        return self

    def detach(self):
        self.F = self.F.detach()
        return self

    def to(self, device, non_blocking=True):
        return self

    def __add__(self, other):
        from .nn import emulate          # (`net(x) + shortcut`: other is the shortcut, models/basic_blocks.py:55)
        t = SparseTensor(self.F + emulate.on_shortcut(other.F), self.C, self.s)
        t.coord_maps = self.coord_maps
        t.kernel_maps = self.kernel_maps
        return t
