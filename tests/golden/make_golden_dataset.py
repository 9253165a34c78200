"""Generate tests/golden/dataset.npz by running the REFERENCE's own `ScannetReferenceDataset.__getitem__`
(lib/dataset.py:64-298, imported from /root/reference in the build container only) on a seeded synthetic scan
written to a temporary directory in the on-disk format the reference loads (`*_aligned_vert.npy`,
`*_ins_label_pg.npy`, `*_sem_label_pg.npy`, `*_aligned_bbox.npy`, lib/dataset.py:94-97).

Shims (none of them touches the arithmetic under test): stub modules for the absent `easydict`, `h5py`, `trimesh`,
`plyfile` (imported at module top by lib/config.py / lib/dataset.py / utils/pc_utils.py, unused on this path),
`yaml.load` given a Loader, argv cleaned, a temporary cwd holding an empty `data/scannet/scans/`, CONF.PATH
repointed (meta data -> the reference's own tsv / npz, scans + glove -> the temporary directory), and `torchsparse`
provided by the oracle restatement (sparse_quantize / SparseTensor are only used for `lidar` / `pts_batch`, which the
fixture stores as order-independent voxel sets). No bytecode is written next to the reference's sources.

The fixture holds inputs' seeds and expected OUTPUTS only. Usage: python tests/golden/make_golden_dataset.py
"""
import os
import pickle
import sys
import tempfile
import types

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), REF, os.path.join(REF, "lib"), os.path.join(REF, "utils")]

from instancerefer_amd import synthetic as S  # noqa: E402

# configuration of the fixture (tests read it from the npz)
CASES = dict(plain=dict(seed=31, augment=False), augmented=dict(seed=32, augment=True))
RAW = dict(num_vertices=9000, num_instances=5, same_class=3)
NUM_POINTS = 6000
SCENE = "scene0000_00"


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install_shims(tmp):
    sys.modules["easydict"] = types.SimpleNamespace(EasyDict=_AttrDict)
    for name in ("h5py", "trimesh", "plyfile"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    import yaml
    orig = yaml.load
    yaml.load = lambda f, Loader=None: orig(f, Loader=yaml.SafeLoader)
    sys.argv = [sys.argv[0], "--config", os.path.join(REF, "config", "InstanceRefer.yaml")]
    os.makedirs(os.path.join(tmp, "data", "scannet", "scans"), exist_ok=True)
    os.chdir(tmp)


def voxel_set(st):
    """Order-independent form of a SparseTensor: rows sorted by (x, y, z)."""
    c = np.asarray(st.C)[:, :3].astype(np.int64)
    f = np.asarray(st.F)
    o = np.lexsort((c[:, 2], c[:, 1], c[:, 0]))
    return c[o].astype(np.int32), f[o]


def main():
    tmp = tempfile.mkdtemp(prefix="irx_golden_")
    install_shims(tmp)
    from lib.config import CONF
    CONF.PATH.SCANNET = os.path.join(REF, "data", "scannet")
    CONF.PATH.SCANNET_META = os.path.join(CONF.PATH.SCANNET, "meta_data")
    CONF.PATH.SCANNET_DATA = os.path.join(tmp, "pointgroup_data")
    CONF.PATH.DATA = tmp
    os.makedirs(CONF.PATH.SCANNET_DATA)
    tokens = ["the", "chair", "is", "next", " ", "to", "the", "brown", "table", "."]
    rng = np.random.default_rng(1)
    # SORTED vocabulary: iterating a Python set of strings depends on PYTHONHASHSEED, which made `lang_feat_rows` differ on
    # every regeneration. "table" and "." are out of vocabulary (-> glove["unk"]), " " exercises isspace().
    glove = {t: rng.standard_normal(300) for t in sorted(set(tokens[:-2]) - {" "}) + ["unk"]}
    with open(os.path.join(tmp, "glove.p"), "wb") as f:
        pickle.dump(glove, f)
    import lib.dataset as D                                  # the reference's dataset module
    DC = D.DC

    out = {"nyu40ids": np.asarray(DC.nyu40ids),
           "nyu40id2class": np.asarray([DC.nyu40id2class.get(i, -1) for i in range(41)]),
           "mean_size_arr": np.asarray(DC.mean_size_arr),
           "raw": np.asarray([RAW["num_vertices"], RAW["num_instances"], RAW["same_class"], NUM_POINTS])}
    for case, cfg in CASES.items():
        raw = S.make_raw_scene(cfg["seed"], **RAW)
        base = os.path.join(CONF.PATH.SCANNET_DATA, SCENE)
        np.save(base + "_aligned_vert.npy", raw["mesh_vertices"])
        np.save(base + "_ins_label_pg.npy", raw["instance_labels"])
        np.save(base + "_sem_label_pg.npy", raw["semantic_labels"])
        np.save(base + "_aligned_bbox.npy", raw["instance_bboxes"])
        scanrefer = [dict(scene_id=SCENE, object_id="1", object_name="chair", ann_id="0", token=tokens)]
        args = S.default_args(num_points=NUM_POINTS, use_augment=cfg["augment"])
        ds = D.ScannetReferenceDataset(scanrefer, [SCENE], split="train", args=args)
        np.random.seed(cfg["seed"])
        torch.manual_seed(cfg["seed"])
        dd = ds[0]
        o = {"seed": np.asarray(cfg["seed"]), "augment": np.asarray(int(cfg["augment"]))}
        for k in ("point_min", "point_max", "point_clouds", "instance_labels", "lang_len", "object_cat", "object_id",
                  "ref_center_label", "ref_size_residual_label", "ref_size_class_label", "ref_heading_class_label",
                  "ref_heading_residual_label", "unique_multiple", "ref_box_label", "center_label",
                  "size_residual_label", "size_class_label", "num_bbox"):
            o[k] = np.asarray(dd[k])
        o["lang_feat_rows"] = np.asarray(dd["lang_feat"][:len(tokens) + 1])
        o["glove_vocab"] = np.asarray(sorted(glove))
        o["glove_vectors"] = np.stack([glove[t] for t in sorted(glove)], 0)
        o["tokens"] = np.asarray(tokens)
        o["instance_points"] = np.stack(dd["instance_points"], 0)
        o["instance_obbs"] = np.stack(dd["instance_obbs"], 0)
        o["instance_class"] = np.asarray(dd["instance_class"])
        o["pred_obb_batch"] = np.asarray(dd["pred_obb_batch"])
        o["lidar_C"], o["lidar_F"] = voxel_set(dd["lidar"])
        o["pts_batch_sizes"] = np.asarray([np.asarray(t.C).shape[0] for t in dd["pts_batch"]])
        c0, f0 = voxel_set(dd["pts_batch"][0])
        o["pts_batch0_C"], o["pts_batch0_F"] = c0, f0
        for k, v in o.items():
            out[case + "/" + k] = v
        print(case, "instances", o["instance_points"].shape, "classes", o["instance_class"], "lidar", o["lidar_C"].shape,
              "pts_batch", o["pts_batch_sizes"], "ref_center", o["ref_center_label"])
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("dataset.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "dataset.npz")), "bytes on disk")


if __name__ == "__main__":
    main()
