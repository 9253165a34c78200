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


def kink_free_state_dict(sd, shift=6.0):
    """A state dict with `shift` added to every BatchNorm shift of the two sparse encoders (attribute.net.*, scene.net.*): their
    ReLU inputs then sit ~6 sigma above zero, so no activation is within fp32 round-off of its kink and two correct fp32
    implementations take the same branch everywhere — which is what makes an ELEMENT-WISE gradient comparison meaningful
    (a single flipped ReLU moves every shallower gradient by ~2e-3)."""
    out = dict(sd)
    for k in out:
        if k.startswith(("attribute.net.", "scene.net.")) and k.endswith(("net.1.bias", "net.4.bias")):
            out[k] = out[k] + shift
    return out


# Parameters whose gradient is MATHEMATICALLY zero: a Linear / Conv2d bias that feeds a train-mode BatchNorm (the batch mean
# removes any per-channel shift: reference models/attribute_module.py:31, relation_module.py:26, scene_module.py:36,42) and the
# biases of the attention logits (softmax is shift invariant: models/lang_module.py:61-83). Both implementations return pure
# cancellation noise there (~1e-5 from sums of O(1) terms), so the bar cannot be relative to the entry. The rule only applies to a
# listed name whose ORACLE gradient is itself below zero_bar * top (a Linear bias in front of a LayerNorm is not zero and is
# compared like every other parameter).
ZERO_GRAD_SUFFIXES = ("lang_emb_fc.0.bias", "vis_emb_fc.0.bias", "vis_emb_fc1.0.bias", "cls.0.bias", "fc_a.bias", "fc_cls.bias",
                      "fc_rel.bias", "fc_scene.bias", "lang_fc.bias", "conv1.0.bias", "conv1.3.bias")


def elementwise_grad_report(named_got, named_exp, rel=1e-3, floor=1e-6, zero_suffixes=ZERO_GRAD_SUFFIXES, zero_bar=1e-4):
    """Per parameter: max |got - exp| against rel * max(max|exp|, floor * top) with top = the largest entry of any gradient;
    parameters named in zero_suffixes (mathematically zero gradients) must stay below zero_bar * top on both sides instead.
    Returns (bad: {name: (err, max|exp|)}, worst ratio err / bar over the ordinary parameters, report lines)."""
    top = max(float(p.grad.abs().max()) for p in named_exp.values() if p.grad is not None)
    bad, worst, lines = {}, 0.0, []
    for n, p in named_exp.items():
        g = named_got[n].grad
        if p.grad is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        assert g is not None and g.shape == p.grad.shape, n
        e = float((g.detach().cpu().double() - p.grad.double()).abs().max())
        m = float(p.grad.abs().max())
        if n.endswith(zero_suffixes) and m <= zero_bar * top:
            mg = float(g.abs().max())
            lines.append("%-52s zero-gradient: |exp| %.1e |got| %.1e (top %.1e)" % (n, m, mg, top))
            if not mg <= zero_bar * top:
                bad[n] = (mg, m)
            continue
        bar = rel * max(m, floor * top)
        worst = max(worst, e / bar)
        lines.append("%-52s err %.1e  max|exp| %.1e  err/bar %.3f" % (n, e, m, e / bar))
        if not e <= bar:
            bad[n] = (e, m)
    return bad, worst, lines
