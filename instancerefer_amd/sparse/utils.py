"""torchsparse.utils surface: sparse_quantize / sparse_collate / sparse_collate_tensors /
sparse_collate_fn (reference call sites: models/attribute_module.py:65-70,101;
lib/dataset.py:229-234,256-261,458), plus the batched device voxeliser the drop-in modules use."""
from collections.abc import Sequence

import numpy as np
import torch

from . import functional as F_
from .tensor import DownMap, Level, SparseTensor


def _voxel3(quantization_size):
    if isinstance(quantization_size, (Sequence, np.ndarray, torch.Tensor)):
        v = [float(q) for q in quantization_size]
        assert len(v) == 3
        return v
    return [float(quantization_size)] * 3


def voxelize(xyz, feats, batch, voxel_size, batch_size):
    """Device voxeliser: one representative (FIRST-occurrence) point per voxel, features NOT averaged —
    torchsparse.sparse_quantize semantics — for a whole batch of clouds in one pass.

    xyz (N,3) f32/f64 cuda; feats (N,C) cuda; batch (N,) int32 cuda or None; returns a canonical
    SparseTensor (rows in Morton order) at stride 1.
    """
    _, keys = F_.quantize(xyz, batch, _voxel3(voxel_size), want_coords=False)
    win = F_.voxel_unique(keys)
    wkeys = keys.index_select(0, win)
    skeys, order = F_.sort_keys(wkeys, F_.morton_bits(batch_size))
    idx = win.index_select(0, order)
    C = F_.keys_to_coords(skeys)          # a key holds its voxel: no gather of coordinate rows through the permutation
    Fv = feats.index_select(0, idx).float().contiguous()
    lv = Level(C, skeys, 1, batch_size)
    return SparseTensor(Fv, C, 1, batch_size, lv)


class VoxelizePending:
    """voxelize() + a `levels`-deep coordinate pyramid with every kernel enqueued and NO host sync yet: buffers are
    sized for the upper bound (one voxel per point), the voxel count travels on the device into irx_pyramid_build and
    arrives on the host together with the level sizes. finish() -> canonical SparseTensor with its pyramid built."""

    def __init__(self, C, Fv, skeys, batch_size, pyramid):
        self.C, self.Fv, self.skeys, self.batch_size, self.pyramid = C, Fv, skeys, batch_size, pyramid

    def finish(self):
        n0, levels = self.pyramid.finish()
        C, Fv, skeys = self.C[:n0], self.Fv[:n0], self.skeys[:n0]
        lv0 = Level(C, skeys, 1, self.batch_size)
        lv = lv0
        for parent, koff, oc, ok, child, ld, m in levels:
            out = Level(oc, ok, lv.stride * 2, lv.batch_size)
            lv._down = DownMap(parent, koff, child, ld, out)
            lv = out
        return SparseTensor(Fv, C, 1, self.batch_size, lv0)




def voxelize_launch(xyz, feats, batch, voxel_size, batch_size, levels):
    """Sync-free voxelize + pyramid (see VoxelizePending). Same result as voxelize() followed by build_pyramid()."""
    _, keys = F_.quantize(xyz, batch, _voxel3(voxel_size), want_coords=False)
    n = keys.shape[0]
    win, count = F_.voxel_unique_launch(keys)                    # winners first, zeros behind; count on the device
    # rows behind the device-side count are padding: the sort reads the count itself and gives them the key
    # `batch_size << 48`, which sorts behind every real Morton key (batch index < batch_size)
    skeys, order = F_.sort_keys(keys.index_select(0, win), F_.morton_bits(batch_size, True), n_dev=count,
                                pad=int(batch_size) << 48)
    idx = win.index_select(0, order)
    C = F_.keys_to_coords(skeys)          # (rows behind the count decode the padding key: sliced off in finish())
    Fv = feats.index_select(0, idx).float().contiguous()
    pyr = F_.pyramid_launch(skeys, C, 1, levels, n0_dev=count)
    return VoxelizePending(C, Fv, skeys, batch_size, pyr)


def sparse_quantize(coords, feats=None, labels=None, ignore_label=255, return_index=False,
                    return_invs=False, hash_type='fnv', quantization_size=1):
    """torchsparse.utils.sparse_quantize on the GPU. numpy / torch in; (coords, feats) out as device
    tensors: coords float64-floored int32 (M,3), feats (M,C) of the first point falling in each voxel.
    Row order is ascending Morton key (torchsparse: ascending FNV hash) — consumers are order-free.
    """
    if labels is not None or return_invs:
        raise NotImplementedError("irx sparse_quantize: labels / inverse maps are not on the hot path")
    dev = torch.device("cuda", torch.cuda.current_device())
    xyz = torch.as_tensor(coords).to(dev)
    if xyz.dtype not in (torch.float32, torch.float64):
        xyz = xyz.double()
    c4, keys = F_.quantize(xyz, None, _voxel3(quantization_size))
    win = F_.voxel_unique(keys)
    wkeys = keys.index_select(0, win)
    _, order = F_.sort_keys(wkeys, F_.morton_bits(1))
    idx = win.index_select(0, order)
    if return_index or feats is None:
        return idx
    f = torch.as_tensor(feats).to(dev)
    return c4.index_select(0, idx)[:, :3].contiguous(), f.index_select(0, idx)


def sparse_collate(coords, feats, labels=None, is_double=False, coord_float=False):
    """Concatenate per-sample (coords, feats) adding the batch index as the 4th coord column."""
    cs, fs = [], []
    for b, (c, f) in enumerate(zip(coords, feats)):
        c = torch.as_tensor(c)
        f = torch.as_tensor(f)
        c = c.float() if coord_float else c.int()
        f = f.double() if is_double else f.float()
        bcol = torch.full((c.shape[0], 1), b, dtype=c.dtype, device=c.device)
        cs.append(torch.cat([c[:, :3], bcol], 1))
        fs.append(f)
    if not cs:
        return torch.zeros((0, 4), dtype=torch.int32), torch.zeros((0, 0))
    return torch.cat(cs, 0), torch.cat(fs, 0)


def sparse_collate_tensors(sparse_tensors):
    if len(sparse_tensors) == 0:
        raise ValueError("sparse_collate_tensors: empty list")
    coords, feats = sparse_collate([x.C for x in sparse_tensors], [x.F for x in sparse_tensors])
    return SparseTensor(feats, coords, sparse_tensors[0].s, batch_size=len(sparse_tensors))


def sparse_collate_fn(batch):
    """dict collate: ndarray -> stacked tensor, SparseTensor -> batched SparseTensor, rest -> list."""
    if isinstance(batch[0], dict):
        out = {}
        for name in batch[0].keys():
            v0 = batch[0][name]
            if isinstance(v0, dict):
                out[name] = sparse_collate_fn([s[name] for s in batch])
            elif isinstance(v0, np.ndarray):
                out[name] = torch.stack([torch.from_numpy(s[name]) for s in batch], 0)
            elif isinstance(v0, torch.Tensor):
                out[name] = torch.stack([s[name] for s in batch], 0)
            elif isinstance(v0, SparseTensor):
                out[name] = sparse_collate_tensors([s[name] for s in batch])
            else:
                out[name] = [s[name] for s in batch]
        return out
    return {"input": sparse_collate_tensors(list(batch))}
