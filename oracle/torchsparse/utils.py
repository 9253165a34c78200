"""torchsparse.utils restated (v1.2): sparse_quantize / sparse_collate / sparse_collate_tensors /
sparse_collate_fn. Reference call sites: models/attribute_module.py:65-70,101;
lib/dataset.py:229-234,256-261,458. Semantics per SURVEY.md App. B.1-B.2."""
from collections.abc import Sequence

import numpy as np
import torch

from .tensor import SparseTensor


EXACT_VOXEL_SET = True       # see sparse_quantize
FNV_COLLISION_EVENTS = 0


def fnv_hash_vec(arr):
    """FNV-1a style 64-bit hash over the integer coordinate columns."""
    assert arr.ndim == 2
    arr = arr.copy()
    with np.errstate(invalid="ignore", over="ignore"):
        # negative floats -> uint64 wraps modulo 2^64 (the x86 behaviour upstream relies on)
        arr = arr.astype(np.int64).astype(np.uint64, copy=False)
        hashed = np.uint64(14695981039346656037) * np.ones(arr.shape[0], dtype=np.uint64)
        for j in range(arr.shape[1]):
            hashed *= np.uint64(1099511628211)
            hashed = np.bitwise_xor(hashed, arr[:, j])
    return hashed


def ravel_hash_vec(arr):
    assert arr.ndim == 2
    arr = arr.copy()
    arr -= arr.min(0)
    arr = arr.astype(np.uint64, copy=False)
    arr_max = arr.max(0).astype(np.uint64) + 1
    keys = np.zeros(arr.shape[0], dtype=np.uint64)
    for j in range(arr.shape[1] - 1):
        keys += arr[:, j]
        keys *= arr_max[j + 1]
    keys += arr[:, -1]
    return keys


def sparse_quantize(coords, feats=None, labels=None, ignore_label=255, return_index=False,
                    return_invs=False, hash_type='fnv', quantization_size=1):
    """floor(coords / voxel) (float64, no min-shift) -> hash -> np.unique(return_index) ->
    (discrete_coords[inds], feats[inds]): ONE first-occurrence point per voxel, rows in ascending hash."""
    use_label = labels is not None
    use_feat = feats is not None
    if not use_label and not use_feat:
        return_index = True
    assert hash_type in ('ravel', 'fnv')
    assert coords.ndim == 2
    if use_feat:
        assert feats.ndim == 2 and coords.shape[0] == feats.shape[0]
    dimension = coords.shape[1]
    if isinstance(quantization_size, (Sequence, np.ndarray, torch.Tensor)):
        assert len(quantization_size) == dimension
        quantization_size = [float(i) for i in quantization_size]
    elif np.isscalar(quantization_size):
        quantization_size = [float(quantization_size)] * dimension
    else:
        raise ValueError('Not supported type for quantization_size.')
    discrete_coords = np.floor(coords / np.array(quantization_size))
    key = ravel_hash_vec(discrete_coords) if hash_type == 'ravel' else fnv_hash_vec(discrete_coords)
    if EXACT_VOXEL_SET and hash_type != 'ravel':
        # Upstream's FNV on a float->uint64 cast of NEGATIVE coordinates is platform-dependent UB and has
        # structured collisions between mixed-sign voxels, e.g. (-2,-4,z) and (0,2,z) hash alike, silently
        # merging distinct voxels. The intended semantics (and irx's) is the exact voxel set, so when FNV
        # collides the oracle switches to the collision-free ravel key and counts the event.
        rk = ravel_hash_vec(discrete_coords)
        if len(np.unique(key)) != len(np.unique(rk)):
            global FNV_COLLISION_EVENTS
            FNV_COLLISION_EVENTS += 1
            key = rk
    if use_label:
        raise NotImplementedError("label voting is not on the InstanceRefer path")
    _, inds, invs = np.unique(key, return_index=True, return_inverse=True)
    if return_index:
        return (inds, invs) if return_invs else inds
    if use_feat:
        out = (discrete_coords[inds], feats[inds])
    else:
        out = (discrete_coords[inds],)
    return out + (invs,) if return_invs else out


def sparse_collate(coords, feats, labels=None, is_double=False, coord_float=False):
    coords_batch, feats_batch = [], []
    for batch_id, _ in enumerate(coords):
        c = coords[batch_id]
        f = feats[batch_id]
        c = torch.from_numpy(np.asarray(c)) if not isinstance(c, torch.Tensor) else c
        f = torch.from_numpy(np.asarray(f)) if not isinstance(f, torch.Tensor) else f
        c = c.float() if coord_float else c.int()
        f = f.double() if is_double else f.float()
        n = c.shape[0]
        bcol = torch.ones(n, 1, dtype=c.dtype) * batch_id
        coords_batch.append(torch.cat((c, bcol), 1))
        feats_batch.append(f)
    return torch.cat(coords_batch, 0), torch.cat(feats_batch, 0)


def sparse_collate_tensors(sparse_tensors):
    coords, feats = sparse_collate([x.C for x in sparse_tensors], [x.F for x in sparse_tensors])
    return SparseTensor(feats, coords, sparse_tensors[0].s)


def sparse_collate_fn(batch):
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
