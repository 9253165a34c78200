"""CPU tests against fixtures produced by the REFERENCE's own code (tests/golden/make_golden.py):
LangModule and get_loss of the drop-in package are plain PyTorch/numpy host code, so they are checked
here without a GPU. Tolerances: fp32 round-off (1e-5 abs unless stated)."""
import os

import numpy as np
import torch

from instancerefer_amd import synthetic as S
from instancerefer_amd.lang_module import LangModule
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from helpers import WEIGHT_SEED

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lang_module_matches_reference():
    gold = np.load(os.path.join(G, "lang.npz"))
    lm = LangModule(18, True, True, 300, 128)
    lm.load_state_dict(S.seeded_state_dict(lm, WEIGHT_SEED + 1))
    lm.eval()
    rng = np.random.default_rng(77)
    lens = np.array([30, 7, 126, 1, 64])
    feat = np.zeros((5, 126, 300), np.float32)
    for i, L in enumerate(lens):
        feat[i, :L] = rng.standard_normal((L, 300)).astype(np.float32) * 0.4
    with torch.no_grad():
        dd = lm({"lang_feat": torch.from_numpy(feat), "lang_len": torch.from_numpy(lens)})
    for k in gold.files:
        assert np.abs(dd[k].numpy() - gold[k]).max() <= 1e-5, k


def test_get_loss_matches_reference():
    gold = np.load(os.path.join(G, "loss.npz"))
    cands = gold["cands"].tolist()
    obbs = gold["pred_obbs"]
    pob, o = [], 0
    for c in cands:
        pob.append(obbs[o:o + c] if c else np.asarray([]))
        o += c
    dd = {k: torch.from_numpy(gold[k]) for k in ("lang_scores", "seg_scores", "object_cat", "point_min", "point_max",
                                                 "ref_center_label", "ref_size_residual_label", "ref_size_class_label",
                                                 "ref_heading_class_label", "ref_heading_residual_label")}
    sc = {}
    for k in ("attribute_scores", "relation_scores", "scene_scores"):
        sc[k] = torch.from_numpy(gold[k].copy()).requires_grad_(True)
        dd[k] = sc[k]
    dd["pred_obb_batch"] = pob
    dd = get_loss(dd, DatasetConfig())
    for k in ("loss", "ref_loss", "lang_loss", "seg_loss", "seg_acc"):
        assert np.abs(dd[k].detach().numpy() - gold["out/" + k]).max() <= 1e-5, k
    lab = np.concatenate([c.numpy() if len(c) else np.zeros(0) for c in dd["cluster_label"]])
    assert np.array_equal(lab, gold["out/cluster_label"])
    dd["loss"].backward()
    for k, v in sc.items():
        assert np.abs(v.grad.numpy() - gold["grad/" + k]).max() <= 1e-5, k


def test_state_dict_layout_matches_reference_checkpoints():
    """Key names a reference checkpoint would carry exist with the expected shapes."""
    from instancerefer_amd.instancerefer import InstanceRefer
    m = InstanceRefer(7, S.default_args())
    sd = m.state_dict()
    assert sd["attribute.net.stem.0.net.0.kernel"].shape == (27, 7, 32)
    assert sd["attribute.net.stage1.0.net.0.kernel"].shape == (8, 32, 64)
    assert sd["attribute.net.stage4.1.net.3.kernel"].shape == (27, 128, 128)
    assert sd["scene.to_bev.1.kernel"].shape == (5, 128, 128)
    assert sd["scene.to_bev.2.running_mean"].shape == (128,)
    assert sd["relation.gcn.mlp.0.weight"].shape == (128, 75)
    assert sd["relation.gcn.weight.2.weight"].shape == (25, 64)
    assert "lang.gru.weight_ih_l0_reverse" in sd
    assert sum(p.numel() for p in m.parameters()) == 8021176
    gold = np.load(os.path.join(G, "model.npz"))
    names = {k[len("grad_norm/"):] for k in gold.files if k.startswith("grad_norm/")}
    mine = {n for n, _ in m.named_parameters()}
    assert names <= mine, sorted(names - mine)[:5]


def test_get_eval_matches_reference():
    """Drop-in get_eval vs the outputs of the reference's lib/eval_helper.get_eval on the same dict (loss.npz)."""
    from instancerefer_amd.eval_helper import get_eval
    gold = np.load(os.path.join(G, "loss.npz"))
    cands = gold["cands"].tolist()
    obbs = gold["pred_obbs"]
    pob, o = [], 0
    for c in cands:
        pob.append(obbs[o:o + c] if c else np.asarray([]))
        o += c
    dd = {k: torch.from_numpy(gold[k]) for k in ("lang_scores", "seg_scores", "object_cat", "point_min", "point_max",
                                                 "ref_center_label", "ref_size_residual_label", "ref_size_class_label",
                                                 "ref_heading_class_label", "ref_heading_residual_label",
                                                 "unique_multiple", "attribute_scores", "relation_scores", "scene_scores")}
    dd["pred_obb_batch"] = pob
    dd = get_eval(get_loss(dd, DatasetConfig()), DatasetConfig())
    assert np.array_equal(np.asarray(dd["ref_acc"]), gold["eval/ref_acc"])
    assert np.abs(np.asarray(dd["ref_iou"]) - gold["eval/ref_iou"]).max() <= 1e-12
    assert np.allclose([dd["ref_iou_rate_0.25"], dd["ref_iou_rate_0.5"], float(dd["lang_acc"])], gold["eval/rates"])
    assert np.array_equal(np.asarray([dd["ref_multiple_mask"], dd["ref_others_mask"]]), gold["eval/masks"])
    assert np.abs(np.asarray(dd["pred_bboxes"]) - gold["eval/pred_bboxes"]).max() <= 1e-12
    assert np.abs(np.asarray(dd["gt_bboxes"]) - gold["eval/gt_bboxes"]).max() <= 1e-12
