"""Shared test helpers: seeded synthetic clouds and row alignment between the product's Morton row
order and the oracle's (torchsparse hash) row order."""
import numpy as np
import torch


# configuration of tests/golden/model.npz (see tests/golden/make_golden.py)
GOLDEN_CFG = dict(batch_size=3, seed=123, num_points=6000, num_instances=6, num_candidates=[4, 1, 3],
                  tokens=[30, 12, 21], points_per_instance=256, variant="corner")
WEIGHT_SEED = 2024


def pack_coords(c):
    c = np.asarray(c).astype(np.int64)
    R, OFF = 1 << 17, 1 << 16
    return ((c[:, 3] * R + (c[:, 0] + OFF)) * R + (c[:, 1] + OFF)) * R + (c[:, 2] + OFF)


def align(coords_a, coords_b):
    """-> (ia, ib) index arrays such that coords_a[ia] == coords_b[ib] row for row (asserts same set)."""
    ka, kb = pack_coords(coords_a), pack_coords(coords_b)
    ia, ib = np.argsort(ka, kind="stable"), np.argsort(kb, kind="stable")
    assert len(ka) == len(kb), "different voxel counts: %d vs %d" % (len(ka), len(kb))
    assert np.array_equal(ka[ia], kb[ib]), "coordinate sets differ"
    return ia, ib


def surface_cloud(rng, n, centre=(0.0, 0.0, 0.0), size=(0.8, 0.6, 0.9), c_extra=4):
    """Points on the faces of an axis-aligned box (what a scanned object looks like) + random features."""
    size = np.asarray(size, dtype=np.float64)
    p = rng.uniform(-0.5, 0.5, (n, 3))
    face = rng.integers(0, 6, n)
    ax = face % 3
    p[np.arange(n), ax] = np.where(face < 3, -0.5, 0.5)
    xyz = p * size + np.asarray(centre)
    feats = rng.uniform(-1, 1, (n, c_extra))
    return np.concatenate([xyz, feats], 1)


def oracle_batch(clouds, voxel):
    """Voxelise each cloud with the oracle's sparse_quantize and collate -> oracle SparseTensor."""
    from oracle.torchsparse import SparseTensor
    from oracle.torchsparse.utils import sparse_quantize, sparse_collate_tensors
    ts = []
    for pc in clouds:
        c, f = sparse_quantize(pc[:, :3], pc, quantization_size=np.array([voxel] * 3))
        ts.append(SparseTensor(f, c))
    return sparse_collate_tensors(ts)


def device_batch(clouds, voxel):
    """The same clouds through the product's GPU voxeliser -> canonical irx SparseTensor."""
    from instancerefer_amd.sparse.utils import voxelize
    dev = torch.device("cuda")
    xyz = torch.from_numpy(np.concatenate([pc[:, :3] for pc in clouds], 0)).to(dev)  # float64
    feats = torch.from_numpy(np.concatenate(clouds, 0)).float().to(dev)
    batch = torch.from_numpy(np.concatenate([np.full(len(pc), i, np.int32) for i, pc in enumerate(clouds)])).to(dev)
    return voxelize(xyz, feats, batch, voxel, len(clouds))
