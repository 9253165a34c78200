"""torchsparse.nn.functional restated on CPU: kernel offsets, coordinate down-sampling, kernel-map
construction (hash + query), per-offset gather -> GEMM -> scatter-add convolution (autograd through
plain torch ops), global max pooling. Semantics per SURVEY.md App. B.3-B.4; this is the algorithm
torchsparse's CPU backend runs (hash table build, per-offset gather/matmul/scatter-add)."""
import copy

import numpy as np
import torch

from ..tensor import SparseTensor
from . import emulate

_R = 1 << 17  # coordinate range for packing (x,y,z,b) into one int64 key
_OFF = 1 << 16


def kernel_offsets(kernel_size, tensor_stride=1, dilation=1):
    """KernelRegion.get_kernel_offset: odd sizes enumerate x fastest over {-(k//2)..k//2}*stride, even
    sizes enumerate z fastest over {0..k-1}*stride."""
    ts, d = tensor_stride, dilation
    if kernel_size % 2 == 1:
        r = (np.arange(-kernel_size // 2 + 1, kernel_size // 2 + 1) * ts * d).tolist()
        return np.array([[x, y, z] for z in r for y in r for x in r], dtype=np.int64)
    r = (np.arange(0, kernel_size) * ts * d).tolist()
    return np.array([[x, y, z] for x in r for y in r for z in r], dtype=np.int64)


def _pack(c):
    c = c.astype(np.int64)
    return ((c[:, 3] * _R + (c[:, 0] + _OFF)) * _R + (c[:, 1] + _OFF)) * _R + (c[:, 2] + _OFF)


def spdownsample(coords, ratio):
    """unique( floor(c / ratio) * ratio , batch ) — FLOOR (Minkowski convention), so negatives round down."""
    c = coords.numpy().astype(np.int64)
    new = np.concatenate([np.floor_divide(c[:, :3], ratio) * ratio, c[:, 3:4]], 1)
    keys = _pack(new)
    _, first = np.unique(keys, return_index=True)
    return torch.from_numpy(new[np.sort(first)].astype(np.int32))


def build_kernel_map(in_coords, out_coords, offsets):
    """For every offset k: pairs (in_idx, out_idx) with in_coord == out_coord + offset_k
    (torchsparse: sphash(out_coords, offsets) queried in the hash table of in_coords)."""
    cin = in_coords.numpy().astype(np.int64)
    cout = out_coords.numpy().astype(np.int64)
    in_keys = _pack(cin)
    order = np.argsort(in_keys, kind="stable")
    sk = in_keys[order]
    maps = []
    for k in range(offsets.shape[0]):
        q = cout.copy()
        q[:, :3] += offsets[k]
        qk = _pack(q)
        pos = np.searchsorted(sk, qk)
        pos_c = np.minimum(pos, len(sk) - 1)
        hit = (sk[pos_c] == qk) if len(sk) else np.zeros(len(qk), bool)
        out_idx = np.nonzero(hit)[0]
        in_idx = order[pos_c[hit]]
        maps.append((torch.from_numpy(in_idx.astype(np.int64)), torch.from_numpy(out_idx.astype(np.int64))))
    return maps


def sparseconv_op(features, kernel, maps, n_out):
    """out[o] += features[i] @ kernel[k] for every pair (i, o) of offset k — gather, GEMM, scatter-add.
    (emulate.MODE set: the same sum with the build's bf16 rounding points, oracle/torchsparse/nn/emulate.py.)"""
    if emulate.MODE is not None:
        return emulate.conv(features, kernel, maps, n_out)
    out = torch.zeros(n_out, kernel.shape[-1], dtype=features.dtype)
    for k, (i_idx, o_idx) in enumerate(maps):
        if i_idx.numel() == 0:
            continue
        out = out.index_add(0, o_idx, features.index_select(0, i_idx).mm(kernel[k]))
    return out


def conv3d(inputs, kernel, kernel_size, bias=None, stride=1, dilation=1, transpose=False):
    features, coords, cur_stride = inputs.F, inputs.C, inputs.s
    assert not transpose, "transposed conv is not on the InstanceRefer path"
    if kernel_size == 1 and stride == 1 and dilation == 1:
        out = features.matmul(kernel)
        if bias is not None:
            out = out + bias
        t = SparseTensor(out, coords, cur_stride)
        t.coord_maps = inputs.coord_maps
        t.kernel_maps = inputs.kernel_maps
        return t
    name = 'k%s_os%d_s%d_d%d' % (kernel_size, cur_stride, stride, dilation)
    offsets = kernel_offsets(kernel_size, cur_stride, dilation)
    if stride > 1:
        new_coords = spdownsample(coords, stride * cur_stride)
        maps = build_kernel_map(coords, new_coords, offsets)
        out = sparseconv_op(features, kernel, maps, new_coords.shape[0])
        if bias is not None:
            out = out + bias
        t = SparseTensor(out, new_coords, cur_stride * stride)
        t.coord_maps = copy.copy(inputs.coord_maps)
        t.check()
        t.kernel_maps = copy.copy(inputs.kernel_maps)
        t.kernel_maps[name] = maps
        return t
    maps = inputs.kernel_maps.get(name, None)
    if maps is None:
        maps = build_kernel_map(coords, coords, offsets)
    out = sparseconv_op(features, kernel, maps, coords.shape[0])
    if bias is not None:
        out = out + bias
    t = SparseTensor(out, coords, cur_stride)
    t.coord_maps = inputs.coord_maps
    t.check()
    t.kernel_maps = copy.copy(inputs.kernel_maps)
    t.kernel_maps[name] = maps
    return t


def global_max_pool(inputs):
    batch_index = inputs.C[:, -1]
    max_index = int(torch.max(batch_index).item())
    outs = []
    for i in range(max_index + 1):
        cur = torch.index_select(inputs.F, 0, torch.where(batch_index == i)[0])
        outs.append(cur.max(0)[0].unsqueeze(0))
    return torch.cat(outs, 0)
