"""get_loss / get_eval with the box arithmetic on the device (csrc/irx_labels.hip: irx_iou_labels, irx_eval_select; SURVEY
§8f rows 2 and 4) against tests/golden/loss.npz — the outputs of the REFERENCE's own lib/loss_helper.get_loss and
lib/eval_helper.get_eval (tests/golden/make_golden.py; includes scenes with 0 and 1 candidates and a max-IoU < 0.2 scene)
— and against the host (numpy float64) path of the same functions, which the device path must equal bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TENSORS = ("lang_scores", "seg_scores", "object_cat", "point_min", "point_max", "ref_center_label",
           "ref_size_residual_label", "ref_size_class_label", "ref_heading_class_label", "ref_heading_residual_label",
           "unique_multiple")


def _batch(gold, dev, with_pack):
    from instancerefer_amd.data import upload_instances
    cands = gold["cands"].tolist()
    obbs = gold["pred_obbs"]
    dd = {k: torch.from_numpy(gold[k]).to(dev) for k in TENSORS}
    dd["_host"] = {k: gold[k] for k in TENSORS}
    sc = {}
    for k in ("attribute_scores", "relation_scores", "scene_scores"):
        sc[k] = torch.from_numpy(gold[k].copy()).to(dev).requires_grad_(True)
        dd[k] = sc[k]
    cat = [int(v) for v in gold["object_cat"]]
    pob, o = [], 0
    for c in cands:
        pob.append(obbs[o:o + c] if c else np.asarray([]))
        o += c
    dd["pred_obb_batch"] = pob
    if with_pack:
        # the boxes as instances of the target class: the resident instance pack the device path reads its boxes from
        dd["instance_points"] = [[np.zeros((1, 7)) for _ in range(c)] for c in cands]
        dd["instance_obbs"] = [[pob[i][j] for j in range(c)] for i, c in enumerate(cands)]
        dd["instance_class"] = [[cat[i]] * c for i, c in enumerate(cands)]
        sel = upload_instances(dd, dev).select(cat)
        assert sel["num_filtered_objs"] == cands
        dd["pred_obb_batch"] = sel["pred_obb_batch"]
        dd["_lang_cls_pred_list"] = cat
    return dd, sc


def test_device_labels_and_eval_match_the_reference_fixture(lib):
    from instancerefer_amd.eval_helper import get_eval
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    gold = np.load(os.path.join(G, "loss.npz"))
    dev = torch.device("cuda")
    out = {}
    for mode in ("device", "host"):
        dd, sc = _batch(gold, dev, with_pack=(mode == "device"))
        dd = get_loss(dd, DatasetConfig())
        assert (dd["_labels"].get("dev") is not None) == (mode == "device")
        for k in ("loss", "ref_loss", "lang_loss", "seg_loss", "seg_acc"):
            assert np.abs(dd[k].detach().cpu().numpy() - gold["out/" + k]).max() <= 1e-5, (mode, k)
        lab = np.concatenate([c.cpu().numpy() if len(c) else np.zeros(0) for c in dd["cluster_label"]])
        assert np.array_equal(lab, gold["out/cluster_label"]), mode
        dd["loss"].backward()
        for k, v in sc.items():
            assert np.abs(v.grad.cpu().numpy() - gold["grad/" + k]).max() <= 1e-5, (mode, k)
        dd = get_eval(dd, DatasetConfig())
        assert np.array_equal(np.asarray(dd["ref_acc"]), gold["eval/ref_acc"]), mode
        assert np.abs(np.asarray(dd["ref_iou"]) - gold["eval/ref_iou"]).max() <= 1e-12, mode
        assert np.allclose([dd["ref_iou_rate_0.25"], dd["ref_iou_rate_0.5"], float(dd["lang_acc"])], gold["eval/rates"])
        assert np.array_equal(np.asarray([dd["ref_multiple_mask"], dd["ref_others_mask"]]), gold["eval/masks"])
        assert np.abs(np.asarray(dd["pred_bboxes"]) - gold["eval/pred_bboxes"]).max() <= 1e-12, mode
        assert np.abs(np.asarray(dd["gt_bboxes"]) - gold["eval/gt_bboxes"]).max() <= 1e-12, mode
        out[mode] = (lab, np.asarray(dd["ref_iou"]), float(dd["ref_loss"].detach()), dd["_labels"]["keep_dev"].cpu().numpy())
    # float64 IoUs, labels and the IoU >= 0.2 gate: the device launch equals the numpy evaluation bit for bit
    for a, b in zip(out["device"], out["host"]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_device_iou_labels_bit_exact_on_random_boxes(lib):
    """1000 scenes of 0..12 random boxes (overlapping, touching, disjoint, degenerate) vs the numpy formulation."""
    from instancerefer_amd import _lib
    from instancerefer_amd.loss_helper import box3d_iou_batch, get_3d_box_batch
    rng = np.random.default_rng(8)
    B = 1000
    counts = rng.integers(0, 13, B)
    total = int(counts.sum())
    obbs = np.zeros((total + 5, 7))
    obbs[:, :3] = rng.uniform(-4, 4, (total + 5, 3))
    obbs[:, 3:6] = rng.uniform(0.05, 3.0, (total + 5, 3))
    obbs[::17, 3:6] = 0.0                                   # degenerate boxes
    gt = np.zeros((B, 7))
    gt[:, :3] = rng.uniform(-4, 4, (B, 3))
    gt[:, 3:6] = rng.uniform(0.05, 3.0, (B, 3))
    filtered = rng.permutation(total + 5)[:total].astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    scene_of = np.repeat(np.arange(B), counts)
    ious = box3d_iou_batch(get_3d_box_batch(obbs[filtered, 3:6], obbs[filtered, 6], obbs[filtered, :3]),
                           get_3d_box_batch(gt[scene_of, 3:6], gt[scene_of, 6], gt[scene_of, :3]))
    exp_lab, exp_best = np.zeros(total, np.float32), np.zeros(B)
    for i in range(B):
        if counts[i]:
            seg = ious[starts[i]:starts[i + 1]]
            exp_lab[starts[i] + int(seg.argmax())] = 1
            exp_best[i] = seg.max()
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    labels = torch.empty(total, dtype=torch.float32, device=dev)
    best = torch.empty(B, dtype=torch.float64, device=dev)
    d_obbs, d_f, d_s, d_gt = t(obbs), t(filtered), t(starts), t(gt)
    _lib.call("irx_iou_labels", _lib.ptr(d_obbs), _lib.ptr(d_f), _lib.ptr(d_s), _lib.ptr(d_gt), B, None, None,
              _lib.ptr(labels), None, None, _lib.ptr(best), _lib.stream_ptr())
    assert np.array_equal(labels.cpu().numpy(), exp_lab)
    assert np.array_equal(best.cpu().numpy(), exp_best)


def test_rotated_boxes_take_the_host_path(lib):
    """the device kernels assume heading 0 (true for every ScanRefer box, lib/dataset.py:216). A batch with a
    rotated candidate box must fall back to the numpy path (which rotates the corners, utils/box_util.py:154-175) and give
    what that path gives without a resident pack."""
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    gold = dict(np.load(os.path.join(G, "loss.npz")))
    gold["pred_obbs"] = gold["pred_obbs"].copy()
    gold["pred_obbs"][1, 6] = 0.7
    dev = torch.device("cuda")
    res = {}
    for mode in ("pack", "host"):
        dd, _ = _batch(gold, dev, with_pack=(mode == "pack"))
        dd = get_loss(dd, DatasetConfig())
        assert dd["_labels"].get("dev") is None, mode
        res[mode] = (np.concatenate([c.cpu().numpy() if len(c) else np.zeros(0) for c in dd["cluster_label"]]),
                     float(dd["ref_loss"].detach()))
    assert np.array_equal(res["pack"][0], res["host"][0]) and res["pack"][1] == res["host"][1]
