"""TEST INFRASTRUCTURE ONLY — numpy restatement of the per-sample input pipeline of the reference,
`ScannetReferenceDataset.__getitem__` (lib/dataset.py:93-298): colour / height features (:99-122), scene
sub-sampling (:124), box labels (:145-151,183-198), augmentation (:154-181, `_translate` :440-453), the instance
loop (bounding box, 1024-point resample, same-class voxelisation; :201-247) and the scene voxelisation (:254-260).
SURVEY.md §8(f) rank 1.

PINNED: checked against tests/golden/dataset.npz, which is the output of the reference's own __getitem__ run in the
build container on a seeded synthetic scan (tests/golden/make_golden_dataset.py), with and without augmentation.
The random numbers are consumed in the reference's order from the same generators (numpy's global RandomState for
the two `random_sampling` sites, torch's default generator for the augmentation draws), so a seeded call reproduces
the reference sample for sample.
"""
import numpy as np
import torch

MEAN_COLOR_RGB = np.array([109.8, 97.2, 83.8])     # lib/dataset.py:22
MAX_NUM_OBJ = 128                                  # lib/dataset.py:21


def point_features(mesh_vertices, use_color=True, use_normal=False, use_height=True, multiview=None):
    """lib/dataset.py:99-122. NOTE the reference normalises the colours IN PLACE on a view of the loaded array, so
    the dtype of `mesh_vertices` (float32 on disk) carries through."""
    v = np.array(mesh_vertices, copy=True)
    if use_color:
        pc = v[:, 0:6]
        pc[:, 3:6] = (pc[:, 3:6] - MEAN_COLOR_RGB) / 256.0
    else:
        pc = v[:, 0:3]
    if use_normal:
        pc = np.concatenate([pc, v[:, 6:9]], 1)
    if multiview is not None:
        pc = np.concatenate([pc, multiview], 1)
    if use_height:
        floor = np.percentile(pc[:, 2], 0.99)
        pc = np.concatenate([pc, np.expand_dims(pc[:, 2] - floor, 1)], 1)
    return pc


def _rot(axis, t):
    c, s = np.cos(t), np.sin(t)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rotate_boxes(boxes, rot, axis):
    """data/scannet/model_util_scannet.py:51-83: centres rotate, the two lengths across the axis become the
    axis-aligned extent of the rotated rectangle."""
    centres, lengths = boxes[:, 0:3], boxes[:, 3:6]
    new_c = np.dot(centres, rot.T)
    a, b = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}[axis]
    d1, d2 = lengths[:, a] / 2.0, lengths[:, b] / 2.0
    e1 = np.zeros((len(d1), 4))
    e2 = np.zeros((len(d1), 4))
    for i, (sa, sb) in enumerate([(-1, -1), (1, -1), (1, 1), (-1, 1)]):
        cr = np.zeros((len(d1), 3))
        cr[:, 0] = sa * d1
        cr[:, 1] = sb * d2
        cr = np.dot(cr, rot.T)
        e1[:, i] = cr[:, 0]
        e2[:, i] = cr[:, 1]
    n1, n2 = 2.0 * e1.max(1), 2.0 * e2.max(1)
    cols = [lengths[:, 0], lengths[:, 1], lengths[:, 2]]
    cols[a], cols[b] = n1, n2
    return np.concatenate([new_c, np.stack(cols, 1)], 1)


def z_rotation(t):
    """utils/pc_utils.py rotz."""
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def get_item(raw, object_id, object_cat, nyu40ids, nyu40id2class, mean_size_arr, num_points=40000, augment=False,
             voxel_size_ap=0.02, voxel_size_glp=0.05, use_color=True, use_normal=False, use_height=True,
             sparse_quantize=None):
    """-> dict with the reference's data_dict entries that depend on the scan (everything but the language keys).
    raw: dict(mesh_vertices, instance_labels, semantic_labels, instance_bboxes). `nyu40id2class`: array indexed by
    nyu40 id (-1 = not a target class). `sparse_quantize`: the voxeliser (oracle.torchsparse.utils.sparse_quantize
    by default)."""
    if sparse_quantize is None:
        from oracle.torchsparse.utils import sparse_quantize
    pc = point_features(raw["mesh_vertices"], use_color, use_normal, use_height)
    ins, sem, boxes = raw["instance_labels"], raw["semantic_labels"], raw["instance_bboxes"]
    choices = np.random.choice(pc.shape[0], num_points, replace=pc.shape[0] < num_points)     # :124
    pc, ins, sem = pc[choices], ins[choices], sem[choices]

    target = np.zeros((MAX_NUM_OBJ, 6))
    size_classes = np.zeros((MAX_NUM_OBJ,))
    size_residuals = np.zeros((MAX_NUM_OBJ, 3))
    nb = min(boxes.shape[0], MAX_NUM_OBJ)
    target[:nb] = boxes[:MAX_NUM_OBJ, 0:6]
    if augment:                                                                                 # :154-181
        if torch.rand(1).item() > 0.5:
            pc[:, 0] = -1 * pc[:, 0]
            target[:, 0] = -1 * target[:, 0]
        if torch.rand(1).item() > 0.5:
            pc[:, 1] = -1 * pc[:, 1]
            target[:, 1] = -1 * target[:, 1]
        for axis in ("x", "y", "z"):
            t = (torch.rand(1).item() * np.pi / 18) - np.pi / 36
            rot = _rot(axis, t) if axis != "z" else z_rotation(t)
            pc[:, 0:3] = np.dot(pc[:, 0:3], rot.T)
            target = rotate_boxes(target, rot, axis)
        shift = (torch.rand(3) - 0.5).tolist()
        xyz = pc[:, :3]
        xyz += shift
        pc[:, :3] = xyz
        target[:, :3] += shift
    cls_of_box = [int(nyu40id2class[int(x)]) for x in boxes[:nb, -2]]
    size_classes[:nb] = cls_of_box
    size_residuals[:nb] = target[:nb, 3:6] - mean_size_arr[cls_of_box]
    ref_box = np.zeros(MAX_NUM_OBJ)
    ref_center, ref_size_cls, ref_size_res = np.zeros(3), 0, np.zeros(3)
    for i, gt in enumerate(boxes[:nb, -1]):
        if gt == object_id:
            ref_box[i] = 1
            ref_center, ref_size_cls, ref_size_res = target[i, 0:3], size_classes[i], size_residuals[i]

    inst_pts, inst_cls, inst_obb, pts_batch, pred_obbs = [], [], [], [], []
    for lab in np.unique(ins):                                                                  # :207-247
        ind = np.nonzero(ins == lab)[0]
        s = sem[ind[0]]
        if s not in nyu40ids:
            continue
        x = pc[ind]
        c = int(nyu40id2class[int(s)])
        inst_cls.append(c)
        lo, hi = x[:, :3].min(0), x[:, :3].max(0)
        obb = np.concatenate((0.5 * (lo + hi), hi - lo, np.array([0])))
        inst_obb.append(obb)
        x = x[np.random.choice(x.shape[0], 1024, replace=x.shape[0] < 1024)]
        inst_pts.append(x)
        if c == object_cat:
            pts_batch.append(sparse_quantize(x[:, :3], x, quantization_size=np.array([voxel_size_ap] * 3)))
            pred_obbs.append(obb)
    lidar = sparse_quantize(pc[:, :3], pc, quantization_size=np.array([voxel_size_glp] * 3))
    return dict(point_clouds=pc.astype(np.float32), instance_labels=ins.astype(np.int64), point_min=pc.min(0)[:3],
                point_max=pc.max(0)[:3], instance_points=inst_pts, instance_class=inst_cls, instance_obbs=inst_obb,
                pts_batch=pts_batch, pred_obb_batch=pred_obbs, lidar=lidar,
                center_label=target.astype(np.float32)[:, 0:3], size_class_label=size_classes.astype(np.int64),
                size_residual_label=size_residuals.astype(np.float32), num_bbox=np.array(nb).astype(np.int64),
                ref_box_label=ref_box.astype(np.int64), ref_center_label=ref_center.astype(np.float32),
                ref_size_class_label=np.array(int(ref_size_cls)).astype(np.int64),
                ref_size_residual_label=ref_size_res.astype(np.float32),
                ref_heading_class_label=np.array(0).astype(np.int64),
                ref_heading_residual_label=np.array(0).astype(np.int64))


def lang_features(tokens, glove, max_len=126):
    """lib/dataset.py:70-92: (embeddings (max_len, 300) float64, lang_len). Written as the reference's loop."""
    emb = np.zeros((max_len, 300))
    for token_id in range(max_len):
        if token_id >= len(tokens):
            break
        token = tokens[token_id]
        if token.isspace():
            continue
        emb[token_id] = glove[token] if token in glove else glove["unk"]
    n = len([t for t in tokens if not t.isspace()])
    return emb, (n if n <= max_len else max_len)
