"""Building blocks — drop-in for the reference's models/basic_blocks.py (same class names, ctor
signatures and parameter names, so reference checkpoints load: `net.0.kernel`, `net.1.weight`, ...).

Differences are all below the API: each Conv3d->BatchNorm(->+res)->ReLU group runs as three HIP
launches (gather-MFMA conv, BN statistics, fused apply) on Morton-ordered voxels; SparseCrop +
ToDenseBEVConvolution collapse into one output-stationary gather over the dense BEV cells (no
(n,128,128) temporary, no scatter-add atomics, no `.item()` sync; reference basic_blocks.py:231-242).
"""
import math

import torch
import torch.nn as nn

from .sparse import SparseTensor
from .sparse import functional as F_
from .sparse import nn as spnn
from .sparse import encoder_fn


class BasicConvolutionBlock(nn.Module):
    """Conv3d -> BatchNorm -> ReLU (reference basic_blocks.py:10-25)."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, transpose=False):
        super().__init__()
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, transpose=transpose),
            spnn.BatchNorm(outc),
            spnn.ReLU(True))

    def forward(self, x):
        return spnn.conv_bn_act(self.net[0], self.net[1], x, relu=True)


class ResidualBlock(nn.Module):
    """relu( BN(conv(relu(BN(conv(x))))) + downsample(x) ) (reference basic_blocks.py:28-56)."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
            spnn.BatchNorm(outc),
            spnn.ReLU(True),
            spnn.Conv3d(outc, outc, kernel_size=ks, dilation=dilation, stride=1),
            spnn.BatchNorm(outc))
        self.downsample = nn.Sequential() if (inc == outc and stride == 1) else \
            nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=1, dilation=1, stride=stride), spnn.BatchNorm(outc))
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        x = x.canonical()
        h = spnn.conv_bn_act(self.net[0], self.net[1], x, relu=True)
        if len(self.downsample) == 0:
            skip = x
        else:
            skip = spnn.conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
        return spnn.conv_bn_act(self.net[3], self.net[4], h, relu=True, residual=skip)


class SparseConvEncoder(nn.Module):
    """13 sparse convs: stem 3^3 C0->32; 4 x {2^3/2 down + Residual(2 x 3^3)} with channels
    64,128,128,128 (reference basic_blocks.py:59-95)."""

    def __init__(self, input_dim):
        super().__init__()
        self.stem = nn.Sequential(BasicConvolutionBlock(input_dim, 32, 3))
        self.stage1 = nn.Sequential(BasicConvolutionBlock(32, 64, ks=2, stride=2), ResidualBlock(64, 64, 3))
        self.stage2 = nn.Sequential(BasicConvolutionBlock(64, 128, ks=2, stride=2), ResidualBlock(128, 128, 3))
        self.stage3 = nn.Sequential(BasicConvolutionBlock(128, 128, ks=2, stride=2), ResidualBlock(128, 128, 3))
        self.stage4 = nn.Sequential(BasicConvolutionBlock(128, 128, ks=2, stride=2), ResidualBlock(128, 128, 3))

    def forward(self, x, defer=False):
        """defer=True (training executor only): issue the pass now and return an encoder_fn.Deferred whose .attach() creates the
        autograd node later — the backward then reaches this encoder in the order of the attach, not of the issue."""
        x = x.canonical()
        x.level().build_pyramid(4)       # all level-size syncs up front, while the GPU queue is still empty
        if encoder_fn.can_fuse(self):
            return encoder_fn.run_encoder(self, x, defer=defer)     # training: the whole encoder as one autograd node
        x = self.stem(x)
        x = self.stage1(x)
        x = self.stage2(x)
        x = self.stage3(x)
        x = self.stage4(x)
        return x


class BEVEncoder(SparseConvEncoder):
    """Identical topology to SparseConvEncoder (reference basic_blocks.py:136-171)."""


class DynamicEdgeConv(nn.Module):
    """Instance-graph edge convolution with max aggregation (reference basic_blocks.py:98-133, which
    derives from torch_geometric MessagePassing(aggr='max') and calls torch_cluster knn).

    message(i <- j) = mlp([x_i, weight([pos_j - pos_i, cls_i, cls_j]), x_j]); out_i = max_j message.
    kNN runs in one HIP launch (one wave per query, batch-segmented) on a fixed (n_query, k) edge grid (slots beyond a
    scene's instance count hold -1), so there is no data-dependent edge count and no host sync; gather, both edge MLPs
    and the masked max are ONE launch forward and one backward (csrc/irx_edgeconv.hip, EdgeConvMaxFn below) instead of
    ~20 + ~40 ATen ops.
    """

    def __init__(self, F_in, F_out, k=6, num_classes=18):
        super().__init__()
        self.k = k
        self.num_classes = num_classes
        self.mlp = nn.Sequential(nn.Linear(3 * F_in, F_out), nn.ReLU(), nn.Linear(F_out, F_out))
        self.weight = nn.Sequential(nn.Linear(3 + num_classes + num_classes, 64), nn.ReLU(), nn.Linear(64, F_in))

    def forward(self, support_xyz, batch_index, filtered_index, features, support_offsets=None, nbr=None):
        if nbr is None:                                  # (nbr: the kNN grid when the input preparation already built it)
            query_xyz = torch.index_select(support_xyz, 0, filtered_index)
            query_batch = torch.index_select(batch_index, 0, filtered_index)
            if support_offsets is None:
                nb = int(batch_index.max().item()) + 1 if batch_index.numel() else 0
                counts = torch.bincount(batch_index, minlength=nb)
                support_offsets = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).int()
            nbr = F_.knn_batched(support_xyz, support_offsets, query_xyz, query_batch.int(), self.k)  # (nq,k)
        if not self.fused_supported(features.shape[1]):
            return self.forward_unfused(support_xyz, filtered_index, features, nbr)
        return EdgeConvMaxFn.apply(features, support_xyz, filtered_index, nbr, self.num_classes,
                                   self.weight[0].weight, self.weight[0].bias, self.weight[2].weight, self.weight[2].bias,
                                   self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias)


    # limits of the fused kernels (csrc/irx_edgeconv.hip: one 16-row MFMA tile of edges per query, the edge-input gradient in a
    # tile of F_out columns); `k` is a YAML knob of the reference (config/InstanceRefer.yaml, models/relation_module.py:13-25)
    FUSED_MAX_K = 16

    def fused_supported(self, f_in):
        hid, f_out = self.weight[0].out_features, self.mlp[2].out_features
        return (self.k <= self.FUSED_MAX_K and hid <= 128 and 3 + 2 * self.num_classes <= f_out and f_in >= self.num_classes
                and self.mlp[0].out_features == f_out)

    def forward_unfused(self, support_xyz, filtered_index, features, nbr):
        """The same function on the (n_query, k) neighbour grid of irx_knn_batched with ATen ops (rocBLAS GEMMs), for graphs
        the fused kernel does not take (k > 16): message(i <- j) = mlp([x_i, weight([pos_j - pos_i, cls_i, cls_j]), x_j])
        (reference models/basic_blocks.py:126-133), max over the valid neighbours, 0 for a query without any (torch_scatter's
        fill, as the fused op)."""
        nq, k = nbr.shape
        nc = self.num_classes
        valid = nbr >= 0
        j = nbr.clamp(min=0).long().view(-1)
        x_i = torch.index_select(features, 0, filtered_index).unsqueeze(1).expand(nq, k, features.shape[1])
        x_j = torch.index_select(features, 0, j).view(nq, k, -1)
        pos_i = torch.index_select(support_xyz, 0, filtered_index).unsqueeze(1)
        pos_j = torch.index_select(support_xyz, 0, j).view(nq, k, 3)
        ew = self.weight(torch.cat([pos_j - pos_i, x_i[..., -nc:], x_j[..., -nc:]], -1))
        msg = self.mlp(torch.cat([x_i, ew, x_j], -1))
        msg = msg.masked_fill(~valid.unsqueeze(-1), float("-inf"))
        out = msg.max(1).values
        return torch.where(valid.any(1, keepdim=True), out, torch.zeros_like(out))


class EdgeConvMaxFn(torch.autograd.Function):
    """out_i = max_j mlp([x_i, weight([pos_j - pos_i, cls_i, cls_j]), x_j]) over the (n_query, k) neighbour grid `nbr`
    (irx_edgeconv_max_fwd / irx_edgeconv_max_bwd, include/irx.h). Gradients: the eight MLP parameters always; the node
    features only when they require it (the reference's features are data); positions never."""

    @staticmethod
    def forward(ctx, feats, pos, qidx, nbr, nc, *params):
        import ctypes
        from . import _lib
        feats, pos = feats.contiguous().float(), pos.contiguous().float()
        qidx, nbr = qidx.contiguous().long(), nbr.contiguous().int()
        params = [p.contiguous().float() for p in params]
        nq, k = nbr.shape
        fin, hid, fout = feats.shape[1], params[0].shape[0], params[6].shape[0]
        out = torch.empty((nq, fout), dtype=torch.float32, device=feats.device)
        arg = torch.empty((nq, fout), dtype=torch.int32, device=feats.device)
        pp = (ctypes.c_void_p * 8)(*[_lib.ptr(p) for p in params])
        _lib.call("irx_edgeconv_max_fwd", _lib.ptr(feats), _lib.ptr(pos), _lib.ptr(qidx), _lib.ptr(nbr), nq, k, fin, nc, hid,
                  fout, pp, _lib.ptr(out), _lib.ptr(arg), _lib.stream_ptr())
        ctx.save_for_backward(feats, pos, qidx, nbr, arg, *params)
        ctx.dims = (nq, k, fin, nc, hid, fout)
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        from . import _lib
        feats, pos, qidx, nbr, arg, *params = ctx.saved_tensors
        nq, k, fin, nc, hid, fout = ctx.dims
        dev = feats.device
        dout = dout.contiguous().float()
        grads = [torch.empty_like(p) for p in params]
        wsb = int(_lib.load().irx_edgeconv_workspace_bytes(nq, k, fin, nc, hid, fout))
        ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
        ein = 3 + 2 * nc
        dmin = torch.empty(nq * k * (3 * fin + ein), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        pp = (ctypes.c_void_p * 8)(*[_lib.ptr(p) for p in params])
        gp = (ctypes.c_void_p * 8)(*[_lib.ptr(g) for g in grads])
        _lib.call("irx_edgeconv_max_bwd", _lib.ptr(feats), _lib.ptr(pos), _lib.ptr(qidx), _lib.ptr(nbr), nq, k, fin, nc, hid,
                  fout, pp, _lib.ptr(dout), _lib.ptr(arg), gp, _lib.ptr(dmin), _lib.ptr(ws), wsb, _lib.stream_ptr())
        dfeats = None
        if dmin is not None:                 # rare (features are data in the reference): scatter the per-edge terms
            dm = dmin[:nq * k * 3 * fin].view(nq, k, 3, fin)
            de = dmin[nq * k * 3 * fin:].view(nq, k, ein)
            j = nbr.clamp(min=0).long().view(-1)
            dfeats = torch.zeros_like(feats)
            dq = dm[:, :, 0].sum(1)
            dq[:, fin - nc:] += de[:, :, 3:3 + nc].sum(1)
            dfeats.index_add_(0, qidx, dq)
            dj = dm[:, :, 2].reshape(-1, fin).clone()
            dj[:, fin - nc:] += de[:, :, 3 + nc:].reshape(-1, nc)
            dfeats.index_add_(0, j, dj)      # rows of missing neighbours carry zeros
        return (dfeats, None, None, None, None) + tuple(grads)


def spcrop(inputs, loc_min, loc_max):
    """Boolean-mask crop (reference basic_blocks.py:174-182). Host-syncing compaction; the scene path does
    not call it — ToDenseBEVConvolution only ever looks up voxels inside its window."""
    x = inputs.canonical()
    coords = x.C
    lo = torch.as_tensor(loc_min, device=coords.device)
    hi = torch.as_tensor(loc_max, device=coords.device)
    valid = ((coords[:, :3] >= lo) & (coords[:, :3] < hi)).all(-1)
    return SparseTensor(x.F[valid], coords[valid].contiguous(), x.s, x._batch_size, None)


class SparseCrop(nn.Module):
    def __init__(self, loc_min, loc_max):
        super().__init__()
        self.loc_min = loc_min
        self.loc_max = loc_max

    def forward(self, inputs):
        return spcrop(inputs, self.loc_min, self.loc_max)


class ToDenseBEVConvolution(nn.Module):
    """Sparse (x,y,z) -> dense BEV (B, C, nx, ny): every voxel is multiplied by the 128x128 kernel of its
    z-bin and voxels sharing an (x,y) cell are summed (reference basic_blocks.py:195-243).

    Here: out[cell] = sum_z F[row(cell, z)] @ kernel[z] — the sparse-conv gather kernel with K = n_z
    "offsets" and the dense cells as output rows. Voxels outside [0,nx)x[0,ny)x[0,nz) (in stride units)
    are never referenced, which is exactly SparseCrop(loc_min=0, loc_max=shape*stride).
    """

    def __init__(self, in_channels, out_channels, shape, offset=(0, 0, 0), z_dim=1, use_bias=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        shape = [int(v) for v in (shape.tolist() if hasattr(shape, "tolist") else shape)]
        off = [int(v) for v in (offset.tolist() if hasattr(offset, "tolist") else offset)]
        if any(off):
            raise NotImplementedError("irx ToDenseBEVConvolution: non-zero offset is not on the InstanceRefer path")
        if use_bias:
            raise NotImplementedError("irx ToDenseBEVConvolution: use_bias is not on the InstanceRefer path")
        if z_dim != 2:
            raise NotImplementedError("irx ToDenseBEVConvolution: z_dim must be 2 (reference scene_module.py:27)")
        self.z_dim = z_dim
        self.n_kernels = shape[z_dim]
        self.bev_dims = [i for i in range(3) if i != z_dim]
        self.bev_shape = [shape[i] for i in self.bev_dims]
        self.kernel = nn.Parameter(torch.zeros(self.n_kernels, in_channels, out_channels))
        self.bias = 0
        self.init_weight()

    def __repr__(self):
        return 'ToDenseBEVConvolution(in_channels=%d, out_channels=%d, n_kernels=%d)' % (
            self.in_channels, self.out_channels, self.n_kernels)

    def init_weight(self):
        std = 1. / math.sqrt(self.in_channels)
        self.kernel.data.uniform_(-std, std)

    def rows(self, inputs):
        """-> (B * nx * ny, Cout) channels-last cell rows (row = b*nx*ny + ix*ny + iy), batch size."""
        x = inputs.canonical()
        lv = x.level()
        nx, ny = self.bev_shape
        nz = self.n_kernels
        tbl, cell, zbin = lv.bev(nx, ny, nz)
        ncell = lv.batch_size * nx * ny

        def tbl_b():
            return F_.kmap_down_transpose(cell, zbin), max(lv.n, 1)

        return F_.SparseConvFn.apply(x.F, self.kernel, tbl, ncell, ncell, tbl_b, 0), lv.batch_size

    def forward(self, inputs, batch_size=None):
        bev, b = self.rows(inputs)
        nx, ny = self.bev_shape
        return bev.view(b, nx, ny, -1).permute(0, 3, 1, 2).contiguous()  # BCHW


# ---- dense 2D head on channels-last cell rows --------------------------------------------------------------
# The scene head (reference scene_module.py:25-38: BatchNorm2d/ReLU, Conv2d 3x3, BatchNorm2d/ReLU, Dropout,
# Conv2d 3x3 on a (B,128,15,25) map) is run on (cells, C) rows with the same HIP kernels as the sparse path:
# a 3x3 "valid" Conv2d is the gather-MFMA conv with a constant 9-offset table over the dense grid, BatchNorm2d
# is BatchNorm over rows. No NCHW<->NHWC permutes, no MIOpen fp32 conv (its solver choice is timing dependent
# and includes Winograd, i.e. run-to-run different numerics). nn.Conv2d / nn.BatchNorm2d modules only hold
# the parameters (state-dict keys unchanged).
_GRID_TABLES = {}


def _grid_tables(b, h, w, ks, device):
    """Constant kernel maps of a ks x ks valid convolution on a dense (b, h, w) grid of rows.
    fwd[k][out_row] = in_row,  bwd[k][in_row] = out_row or -1;  k = di*ks + dj (cross-correlation)."""
    key = (b, h, w, ks, str(device))
    if key not in _GRID_TABLES:
        ho, wo = h - ks + 1, w - ks + 1
        bb, ii, jj = torch.meshgrid(torch.arange(b), torch.arange(ho), torch.arange(wo), indexing="ij")
        fwd = torch.empty((ks * ks, b * ho * wo), dtype=torch.int32)
        bwd = torch.full((ks * ks, b * h * w), -1, dtype=torch.int32)
        out_row = (bb * ho * wo + ii * wo + jj).reshape(-1)
        for di in range(ks):
            for dj in range(ks):
                in_row = (bb * h * w + (ii + di) * w + (jj + dj)).reshape(-1)
                fwd[di * ks + dj] = in_row.int()
                bwd[di * ks + dj, in_row] = out_row.int()
        _GRID_TABLES[key] = (fwd.to(device).contiguous(), bwd.to(device).contiguous(), b * ho * wo, b * h * w)
    return _GRID_TABLES[key]


def conv2d_rows(conv: nn.Conv2d, x, b, h, w):
    """nn.Conv2d (square kernel, stride 1, no padding) applied to channels-last rows x (b*h*w, Cin)."""
    ks = conv.kernel_size[0]
    assert conv.kernel_size == (ks, ks) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
    fwd, bwd, n_out, n_in = _grid_tables(b, h, w, ks, x.device)
    wk = conv.weight.permute(2, 3, 1, 0).reshape(ks * ks, conv.in_channels, conv.out_channels)
    y = F_.SparseConvFn.apply(x, wk, fwd, n_out, n_out, lambda: (bwd, n_in), 0)
    if conv.bias is not None:
        y = y + conv.bias
    return y


def batchnorm_rows(bn, x, relu=False):
    """nn.BatchNorm1d/2d semantics on (rows, C) with the fused statistics + apply(+ReLU) kernels."""
    if bn.training or not bn.track_running_stats:
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        mom = 0.0 if bn.momentum is None else bn.momentum
        return F_.BatchNormActFn.apply(x, bn.weight, bn.bias, None,
                                       bn.running_mean if bn.track_running_stats else None,
                                       bn.running_var if bn.track_running_stats else None, bn.eps, mom, relu,
                                       F_.sync_group() if getattr(bn, "_irx_sync", False) else None)
    return F_.bn_eval(x, bn.weight, bn.bias, None, bn.running_mean, bn.running_var, bn.eps, relu)
