"""Seeded synthetic ScanRefer-shaped batches (shapes from the reference's
lib/dataset.py:64-300 + collate_fn :456-469). Used by bench.py, __graft_entry__.smoke() and the tests —
there is no network for the real dataset.

Scene i of a batch uses numpy Generator(seed + i): room 8 x 10 x 3 m with the origin at a corner (or
centred on the origin with variant='centred', to exercise negative voxel coordinates); half of the P
points on floor + walls, the rest on the faces of I axis-aligned boxes resting on the floor; features
xyz + normalised rgb + height (C0 = 7); every instance resampled to 1024 points; class = id mod 18 except
that the first `c` instances share the target class; a T-token utterance of N(0, 0.4) GloVe-like rows.
`multiview=128` inserts the ENet multiview features of the stress config (lib/dataset.py:112-118: concatenated after the
colours, before the height -> C0 = 135; synthetic: N(0,1) clipped at 0, like post-ReLU activations).
"""
import numpy as np
import torch

MEAN_COLOR_RGB = np.array([109.8, 97.2, 83.8])   # reference lib/dataset.py:22
MAX_DES_LEN = 126                                # reference lib/config.py:74


def _box_surface(rng, n, centre, size):
    p = rng.uniform(-0.5, 0.5, (n, 3))
    face = rng.integers(0, 6, n)
    p[np.arange(n), face % 3] = np.where(face < 3, -0.5, 0.5)
    return p * size + centre


def make_scene(seed, num_points=50000, num_instances=8, num_candidates=4, target_class=4, tokens=30,
               points_per_instance=1024, variant="corner", num_classes=18, multiview=0):
    rng = np.random.default_rng(seed)
    room = np.array([8.0, 10.0, 3.0])
    n_bg = num_points // 2
    n_obj = num_points - n_bg
    # background: floor + 4 walls, area-weighted
    areas = np.array([room[0] * room[1], room[0] * room[2], room[0] * room[2], room[1] * room[2], room[1] * room[2]])
    which = rng.choice(5, n_bg, p=areas / areas.sum())
    bg = rng.uniform(0, 1, (n_bg, 3)) * room
    bg[which == 0, 2] = 0.0
    bg[which == 1, 1] = 0.0
    bg[which == 2, 1] = room[1]
    bg[which == 3, 0] = 0.0
    bg[which == 4, 0] = room[0]
    xyz = [bg]
    ins = [np.zeros(n_bg, np.int64)]
    sizes, centres = [], []
    per = n_obj // num_instances
    for j in range(num_instances):
        size = rng.uniform(0.4, 1.2, 3)
        cxy = rng.uniform([0.8, 0.8], [room[0] - 0.8, room[1] - 0.8])
        centre = np.array([cxy[0], cxy[1], size[2] / 2])
        n_j = per if j < num_instances - 1 else n_obj - per * (num_instances - 1)
        xyz.append(_box_surface(rng, n_j, centre, size))
        ins.append(np.full(n_j, j + 1, np.int64))
    xyz = np.concatenate(xyz, 0).astype(np.float32).astype(np.float64)   # vertices are float32 on disk
    ins = np.concatenate(ins, 0)
    if variant == "centred":
        xyz[:, :2] -= room[:2] / 2
    rgb = (rng.uniform(0, 255, (num_points, 3)) - MEAN_COLOR_RGB) / 256.0
    floor = np.percentile(xyz[:, 2], 0.99)
    height = xyz[:, 2:3] - floor
    cols = [xyz, rgb]
    if multiview:
        cols.append(np.maximum(rng.standard_normal((num_points, multiview)), 0.0))
    pc = np.concatenate(cols + [height], 1)                               # (P, 7 [+ multiview]) float64

    instance_points, instance_obbs, instance_class = [], [], []
    for j in range(num_instances):
        x = pc[ins == j + 1]
        lo, hi = x[:, :3].min(0), x[:, :3].max(0)
        instance_obbs.append(np.concatenate([0.5 * (lo + hi), hi - lo, np.array([0.0])]))
        sel = rng.choice(x.shape[0], points_per_instance, replace=x.shape[0] < points_per_instance)
        instance_points.append(x[sel])
        instance_class.append(target_class if j < num_candidates else (target_class + 1 + j % (num_classes - 1)) % num_classes)
    lang = np.zeros((MAX_DES_LEN, 300), np.float32)
    lang[:tokens] = (rng.standard_normal((tokens, 300)) * 0.4).astype(np.float32)
    gt = instance_obbs[0]
    return dict(
        scene_points=pc, instance_points=instance_points, instance_obbs=instance_obbs,
        instance_class=instance_class, lang_feat=lang, lang_len=np.array(tokens, np.int64),
        object_cat=np.array(target_class, np.int64), point_min=pc.min(0)[:3], point_max=pc.max(0)[:3],
        ref_center_label=gt[:3].astype(np.float32), ref_size_residual_label=(gt[3:6] - 1.0).astype(np.float32),
        ref_size_class_label=np.array(target_class, np.int64), ref_heading_class_label=np.array(0, np.int64),
        ref_heading_residual_label=np.array(0, np.int64),
        unique_multiple=np.array(0 if num_candidates == 1 else 1, np.int64), scan_idx=np.array(seed, np.int64))


# nyu40 ids used by make_raw_scene: wall / floor for the background (not in DC.nyu40ids -> "scene_points"
# branch of lib/dataset.py:246-247), then object ids cycling through real ScanNet categories
_NYU40_OBJECT_IDS = (5, 5, 5, 7, 4, 3, 6, 14, 33, 34, 39, 24)


def make_raw_scene(seed, num_vertices=60000, num_instances=8, same_class=3):
    """What lib/dataset.py:94-97 loads from disk for one scan: `mesh_vertices` (V, 6) float32 [xyz, rgb 0..255],
    `instance_labels` (V,) (0 = none, object_id + 1 otherwise), `semantic_labels` (V,) nyu40 ids,
    `instance_bboxes` (I, 8) float64 [centre, size, nyu40 id, object id]. The first `same_class` objects share
    nyu40 id 5 (chair)."""
    rng = np.random.default_rng(seed)
    room = np.array([8.0, 10.0, 3.0])
    n_bg = num_vertices // 2
    per = (num_vertices - n_bg) // num_instances
    bg = rng.uniform(0, 1, (n_bg, 3)) * room
    which = rng.integers(0, 5, n_bg)
    bg[which == 0, 2] = 0.0
    bg[which == 1, 1] = 0.0
    bg[which == 2, 1] = room[1]
    bg[which == 3, 0] = 0.0
    bg[which == 4, 0] = room[0]
    xyz, ins, sem, boxes = [bg], [np.zeros(n_bg, np.int64)], [np.where(which == 0, 2, 1)], []
    for j in range(num_instances):
        size = rng.uniform(0.4, 1.2, 3)
        cxy = rng.uniform([0.8, 0.8], [room[0] - 0.8, room[1] - 0.8])
        centre = np.array([cxy[0], cxy[1], size[2] / 2])
        n_j = per if j < num_instances - 1 else num_vertices - n_bg - per * (num_instances - 1)
        xyz.append(_box_surface(rng, n_j, centre, size))
        ins.append(np.full(n_j, j + 1, np.int64))
        nyu = 5 if j < same_class else _NYU40_OBJECT_IDS[j % len(_NYU40_OBJECT_IDS)]
        sem.append(np.full(n_j, nyu, np.int64))
        boxes.append(np.concatenate([centre, size, [nyu, j]]))
    xyz = np.concatenate(xyz, 0)
    perm = rng.permutation(num_vertices)                       # scans are not sorted by instance
    verts = np.concatenate([xyz, rng.uniform(0, 255, (num_vertices, 3))], 1).astype(np.float32)[perm]
    return dict(mesh_vertices=verts, instance_labels=np.concatenate(ins)[perm],
                semantic_labels=np.concatenate(sem)[perm], instance_bboxes=np.asarray(boxes, np.float64))


_STACK = ("lang_feat", "lang_len", "object_cat", "point_min", "point_max", "ref_center_label",
          "ref_size_residual_label", "ref_size_class_label", "ref_heading_class_label",
          "ref_heading_residual_label", "unique_multiple", "scan_idx")


def collate(scenes):
    """Reference collate (sparse_collate_fn semantics): ndarray -> stacked tensor, everything else -> list.
    `lidar` is NOT built here: callers voxelise `scene_points` with their own voxeliser."""
    out = {}
    for k in _STACK:
        out[k] = torch.stack([torch.from_numpy(np.asarray(s[k])) for s in scenes], 0)
    for k in ("scene_points", "instance_points", "instance_obbs", "instance_class"):
        out[k] = [s[k] for s in scenes]
    return out


def make_batch(batch_size, seed=123, **scene_kw):
    """Scene i gets seed + i. `num_candidates` / `tokens` may be ints or per-scene lists (e.g. [4, 1, 3])."""
    nc = scene_kw.pop("num_candidates", 4)
    tk = scene_kw.pop("tokens", 30)
    scenes = []
    for i in range(batch_size):
        c = nc[i] if isinstance(nc, (list, tuple)) else nc
        t = tk[i] if isinstance(tk, (list, tuple)) else tk
        scenes.append(make_scene(seed + i, num_candidates=c, tokens=t, **scene_kw))
    return collate(scenes)


def to_device(data_dict, device, voxel_size_glp=0.05):
    """What lib/solver.py:242-245 does (move the tensor keys to the GPU) + build `lidar` by voxelising every
    scene of the batch in one GPU pass + upload the instance pack once."""
    from .data import upload_instances
    from .sparse.utils import voxelize
    # labels originate on the host: keep numpy copies so get_loss needs no D2H round trip
    data_dict["_host"] = {k: data_dict[k].numpy() for k in ("ref_center_label", "ref_size_residual_label",
                                                            "ref_heading_class_label", "ref_heading_residual_label",
                                                            "ref_size_class_label", "object_cat", "point_min",
                                                            "point_max")}
    data_dict["lang_len_max"] = int(data_dict["lang_len"].max())
    for k in ("lang_feat", "lang_len", "object_cat", "point_min", "point_max", "ref_center_label",
              "ref_size_residual_label"):
        data_dict[k] = data_dict[k].to(device)
    pts = [torch.from_numpy(p) for p in data_dict["scene_points"]]
    allp = torch.cat(pts, 0).to(device)
    batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(device)
    data_dict["lidar"] = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [voxel_size_glp] * 3, len(pts))
    upload_instances(data_dict, device)
    return data_dict


def default_args(**over):
    """The hot-path keys of config/InstanceRefer.yaml (reference :1-58) as the flat namespace lib/config.py builds."""
    from types import SimpleNamespace
    a = dict(num_classes=18, use_bidir=True, language_module="lang_module", attribute_module="attribute_module",
             relation_module="relation_module", scene_module="scene_module", voxel_size_ap=0.02,
             voxel_size_glp=0.05, k=8, use_gt_lang=True, use_color=True, use_height=True, use_normal=False,
             use_multiview=False, num_points=40000, batch_size=64)
    a.update(over)
    return SimpleNamespace(**a)


def seeded_state_dict(module, seed):
    """Deterministic weights from numpy's PCG64 (independent of torch's RNG / device): conv kernels and
    dense weights U(+-1/sqrt(fan_in)), norm scales U(.5,1.5), biases / running means U(-.1,.1), running
    variances U(.5,1.5). The golden fixtures are generated with exactly this function."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, t in sorted(module.state_dict().items()):
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            v = np.zeros(shape, np.int64)
        elif name.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            v = rng.uniform(-0.1, 0.1, shape)
        elif t.dim() >= 2:
            fan_in = int(np.prod(shape[:-1])) if name.endswith("kernel") else int(np.prod(shape[1:]))
            b = 1.0 / np.sqrt(max(fan_in, 1))
            v = rng.uniform(-b, b, shape)
        elif name.endswith("weight"):
            v = rng.uniform(0.5, 1.5, shape)
        else:
            v = rng.uniform(-0.1, 0.1, shape)
        sd[name] = torch.from_numpy(np.asarray(v)).to(t.dtype)
    return sd


# ---- multiview frames for the projection step (lib/projection.py; scripts/project_multiview_features.py:28-29) ----
PROJ_INTRINSICS = [[37.01983, 0, 20, 0], [0, 38.52470, 15.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
PROJ_ARGS = dict(depth_min=0.1, depth_max=4.0, image_dims=[41, 32], accuracy=0.05)


def make_frames(seed, points, num_frames=4, channels=8):
    """Seeded camera frames looking into the cloud `points` (N, 3): camera_to_world (F, 4, 4) float32, z-buffer depth
    maps (F, 32, 41) float32 rendered from the points themselves (+ noise, holes and out-of-range pixels, so that every
    rejection branch of compute_projection is taken) and per-frame image features (F, channels, 32, 41) float32."""
    rng = np.random.default_rng(seed)
    pts = np.asarray(points, np.float64)[:, :3]
    lo, hi = pts.min(0), pts.max(0)
    fx, fy, cx, cy = PROJ_INTRINSICS[0][0], PROJ_INTRINSICS[1][1], PROJ_INTRINSICS[0][2], PROJ_INTRINSICS[1][2]
    W, H = PROJ_ARGS["image_dims"]
    poses, depths = [], []
    for _ in range(num_frames):
        eye = np.array([rng.uniform(lo[0], hi[0]), rng.uniform(lo[1], hi[1]), rng.uniform(1.2, 1.8)])
        target = np.array([rng.uniform(lo[0], hi[0]), rng.uniform(lo[1], hi[1]), rng.uniform(0.0, 1.0)])
        fwd = target - eye
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye      # camera looks along +z
        cam = (np.linalg.inv(c2w) @ np.concatenate([pts, np.ones((len(pts), 1))], 1).T)
        z = cam[2]
        ok = z > 0.05
        u = np.round(cam[0, ok] * fx / z[ok] + cx).astype(np.int64)
        v = np.round(cam[1, ok] * fy / z[ok] + cy).astype(np.int64)
        inside = (u >= 0) & (u < W) & (v >= 0) & (v < H)
        depth = np.full((H, W), 10.0)
        np.minimum.at(depth, (v[inside], u[inside]), z[ok][inside])
        depth += rng.normal(0, 0.02, depth.shape)                # some points fail the accuracy test
        depth[rng.uniform(size=depth.shape) < 0.05] = 0.0        # holes in the depth image
        poses.append(c2w.astype(np.float32))
        depths.append(depth.astype(np.float32))
    feats = rng.standard_normal((num_frames, channels, H, W)).astype(np.float32)
    return np.stack(poses), np.stack(depths), feats
