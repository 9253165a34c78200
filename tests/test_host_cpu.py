"""Host logic that needs no GPU: candidate filtering vs the reference rule, collate, synthetic determinism,
Morton key helper, gradient all-reduce over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from instancerefer_amd import synthetic as S
from instancerefer_amd.data import InstancePack


def test_candidate_selection_matches_reference_rule():
    dd = S.make_batch(4, seed=5, num_points=2000, num_instances=6, num_candidates=[3, 1, 0, 2], points_per_instance=16)
    dd["instance_class"][2] = [9] * 6     # scene 2: nothing matches the target class
    pack = InstancePack(dd, torch.device("cpu"))
    cls = dd["object_cat"].tolist()
    sel = pack.select(cls)
    # reference: models/attribute_module.py:49-79 / relation_module.py:50-76
    exp_nfo, exp_cand, exp_support, exp_q = [], [], [], []
    flat = 0
    for i in range(4):
        mine = [j for j, c in enumerate(dd["instance_class"][i]) if c == cls[i]]
        exp_nfo.append(len(mine))
        if len(mine) >= 2:
            base = len(exp_support)
            exp_cand += [flat + j for j in mine]
            exp_support += [flat + j for j in range(6)]
            exp_q += [base + j for j in mine]
        flat += 6
    assert sel["num_filtered_objs"] == exp_nfo == [3, 1, 0, 2]
    assert sel["cand"] == exp_cand and sel["support"] == exp_support and sel["query_in_support"] == exp_q
    assert sel["pred_obb_batch"][2].shape == (0,) and sel["pred_obb_batch"][0].shape == (3, 7)
    assert sel["support_scene_offsets"] == [0, 6, 12]


def test_sparse_collate_matches_oracle():
    from instancerefer_amd.sparse import SparseTensor, utils
    from oracle.torchsparse import SparseTensor as OT
    from oracle.torchsparse.utils import sparse_collate_tensors as oc
    rng = np.random.default_rng(0)
    cs = [np.floor(rng.uniform(-5, 5, (n, 3))) for n in (4, 7, 1)]
    fs = [rng.standard_normal((len(c), 3)) for c in cs]
    a = utils.sparse_collate_tensors([SparseTensor(f, c) for f, c in zip(fs, cs)])
    b = oc([OT(f, c) for f, c in zip(fs, cs)])
    assert torch.equal(a.C, b.C) and torch.equal(a.F, b.F) and a.C.dtype == torch.int32 and a.F.dtype == torch.float32
    assert a.batch_size == 3


def test_synthetic_is_deterministic_and_shaped():
    a = S.make_batch(2, seed=9, num_points=3000, num_instances=4, num_candidates=2)
    b = S.make_batch(2, seed=9, num_points=3000, num_instances=4, num_candidates=2)
    assert np.array_equal(a["scene_points"][1], b["scene_points"][1])
    assert a["lang_feat"].shape == (2, 126, 300) and a["instance_points"][0][0].shape == (1024, 7)
    assert a["scene_points"][0].dtype == np.float64 and a["point_min"].dtype == torch.float64
    c = S.make_batch(1, seed=9, num_points=3000, variant="centred")
    assert c["scene_points"][0][:, 0].min() < 0


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    from instancerefer_amd.ddp import FlatGradAllReduce, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    x = torch.randn(8, 8)
    red = FlatGradAllReduce(model.parameters())
    lo, hi = shard_range(8, rank, world)
    red.zero_grad()
    if rank == 1:
        pass                                  # a rank with an empty shard still joins with zero grads
    else:
        model(x[lo:hi]).mean().backward()
    red.all_reduce()
    out[rank] = red.flat.clone()
    dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_world2():
    mp.set_start_method("spawn", force=True)
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29000 + os.getpid() % 2000
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.equal(out[0], out[1])
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    x = torch.randn(8, 8)
    model(x[0:4]).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]) / 2     # rank 1 contributed zeros
    assert torch.allclose(out[0], ref, atol=1e-7)


@pytest.mark.parametrize("case", ["plain", "augmented"])
def test_input_pipeline_host_half_matches_reference_getitem(case):
    """scene_input.draw_sample (labels, classes, RNG consumption) against the fixture made by the reference's own
    ScannetReferenceDataset.__getitem__ (tests/golden/make_golden_dataset.py). The device half is a GPU test."""
    import os
    from instancerefer_amd import scene_input as SI

    class HostScan:                      # ResidentScan without the upload
        def __init__(self, raw):
            self.instance_labels, self.semantic_labels = raw["instance_labels"], raw["semantic_labels"]
            self.instance_bboxes = raw["instance_bboxes"]
            self.num_vertices = raw["mesh_vertices"].shape[0]

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz"))
    nv, ni, sc, npts = (int(v) for v in g["raw"])
    seed = int(g[case + "/seed"])
    raw = S.make_raw_scene(seed, num_vertices=nv, num_instances=ni, same_class=sc)
    tables = SI.ClassTables(g["nyu40ids"], g["nyu40id2class"], g["mean_size_arr"])
    np.random.seed(seed)
    torch.manual_seed(seed)
    d = SI.draw_sample(HostScan(raw), int(g[case + "/object_id"]), tables, num_points=npts,
                       augment=bool(g[case + "/augment"]))
    for k, v in d.labels.items():
        assert np.array_equal(v, g[case + "/" + k]) and v.dtype == g[case + "/" + k].dtype, k
    assert np.array_equal(d.instance_labels, g[case + "/instance_labels"])
    assert d.classes == list(g[case + "/instance_class"])
    assert d.rows.shape == (len(d.classes), 1024) and d.seg[-1] == len(d.order)
    # the features are static per scan: colour / height columns of the sampled cloud
    pc = SI.point_features(raw["mesh_vertices"])[d.choices]
    assert np.array_equal(pc[:, 3:], g[case + "/point_clouds"][:, 3:])
    if case == "plain":
        assert np.array_equal(pc, g[case + "/point_clouds"])
        assert np.array_equal(pc[d.rows], g[case + "/instance_points"])


@pytest.mark.parametrize("augment", [False, True])
def test_batched_box_labels_equal_the_per_sample_reference_order_code(augment):
    """Device mode computes the augmentation parameters + box labels of a whole batch with vectorised numpy
    (scene_input._augment_and_box_labels_batch); fed the same random numbers it must equal the per-sample code that is
    pinned to the reference's __getitem__ (flips, three rotations of the axis-aligned boxes, shift, size residuals,
    reference-target lookup)."""
    from instancerefer_amd import scene_input as SI
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz"))
    tables = SI.ClassTables(g["nyu40ids"], g["nyu40id2class"], g["mean_size_arr"])

    class HostScan:
        def __init__(self, raw):
            self.instance_bboxes = raw["instance_bboxes"]
    scans = [HostScan(S.make_raw_scene(40 + i, num_vertices=3000, num_instances=4 + i, same_class=2)) for i in range(3)]
    oids = [0, 1, 2]
    torch.manual_seed(3)
    batch = [SI._Draw() for _ in scans]
    SI._augment_and_box_labels_batch(batch, scans, oids, tables, augment)
    torch.manual_seed(3)
    r = torch.rand((3, 8))
    orig = torch.rand
    try:
        for i, (sc, oid) in enumerate(zip(scans, oids)):
            feed = iter([r[i, 0:1], r[i, 1:2], r[i, 2:3], r[i, 3:4], r[i, 4:5], r[i, 5:8]])
            torch.rand = lambda *a, **k: next(feed)
            d = SI._Draw()
            SI._augment_and_box_labels(d, sc, oid, tables, augment)
            for k in d.labels:
                assert np.allclose(d.labels[k], batch[i].labels[k], atol=1e-6), (i, k)
                assert d.labels[k].dtype == batch[i].labels[k].dtype, k
            assert (d.flip_x, d.flip_y) == (batch[i].flip_x, batch[i].flip_y)
            if augment:
                assert np.allclose(np.stack(d.rot), np.stack(batch[i].rot), atol=1e-7)
                assert np.allclose(d.shift, batch[i].shift)
    finally:
        torch.rand = orig
