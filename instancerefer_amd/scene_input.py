"""Device-side per-sample input pipeline: what the reference does in numpy inside
`ScannetReferenceDataset.__getitem__` (lib/dataset.py:93-298) + `collate_fn` (:456-469) for every sample of every step
— scene sub-sampling, augmentation, the instance loop (box, 1024-point resample), both voxelisations — on a scan that
stays RESIDENT in HBM (a ScanNet scan is ~2.4 MB of float32 features; all 562 training scans fit in 1.4 GB, and
ScanRefer revisits each scan ~65 times per epoch).

Split of work:
  host  (`draw_sample`)  integer label logic on the (V,) label arrays and every random draw, consumed from the SAME
                         generators in the SAME order as the reference (numpy global RandomState for the two
                         `random_sampling` sites, torch's default generator for the augmentation), so a seeded run
                         reproduces the reference sample for sample; box labels (<= 128 boxes) in numpy.
  device (`build_batch`) irx_scene_sample (gather + flips / rotations / shift), irx_instance_split (boxes, resample,
                         scene extent), the scene voxelisation; the candidate voxelisation happens in the model's
                         prepare stage from the device-resident InstancePack as before. Only index arrays go up
                         (~0.5 MB per sample instead of ~3 MB of float64 points) and 7 floats per instance come back.

The CPU restatement used as the checker is oracle/dataset_ref.py (pinned to the reference's own __getitem__ by
tests/golden/dataset.npz); this module never imports it.
"""
import numpy as np
import torch

from . import _lib
from .data import InstancePack, idx_tensor

MEAN_COLOR_RGB = np.array([109.8, 97.2, 83.8])     # lib/dataset.py:22
MAX_NUM_OBJ = 128                                  # lib/dataset.py:21
NUM_INSTANCE_POINTS = 1024                         # lib/dataset.py:224
import os as _os
_VOXELIZE_LAUNCH = _os.environ.get("IRX_INPUT_VOXELIZE_LAUNCH", "0") == "1"     # dev switch, see PendingBatch._assemble


class ClassTables:
    """The pieces of data/scannet/model_util_scannet.py:ScannetDatasetConfig this path reads: `nyu40ids` (the
    nyu40 ids that are objects), `nyu40id2class` (array indexed by nyu40 id, -1 elsewhere), `mean_size_arr`."""

    def __init__(self, nyu40ids, nyu40id2class, mean_size_arr):
        self.nyu40ids = np.asarray(nyu40ids)
        self.nyu40id2class = np.asarray(nyu40id2class)
        self.mean_size_arr = np.asarray(mean_size_arr)
        self._is_object = np.zeros(max(int(self.nyu40ids.max()) + 1, len(self.nyu40id2class)), bool)
        self._is_object[self.nyu40ids] = True

    def is_object(self, nyu):
        nyu = int(nyu)
        return 0 <= nyu < len(self._is_object) and bool(self._is_object[nyu])


def point_features(mesh_vertices, use_color=True, use_normal=False, use_height=True, multiview=None):
    """Static per scan (lib/dataset.py:99-122): normalised colours, optional normals / multiview features, height above
    the 0.99-percentile floor. dtype follows `mesh_vertices` (the reference works in place on the loaded array)."""
    v = np.array(mesh_vertices, copy=True)
    cols = v[:, 0:6] if use_color else v[:, 0:3]
    if use_color:
        cols[:, 3:6] = (cols[:, 3:6] - MEAN_COLOR_RGB) / 256.0
    if use_normal:
        cols = np.concatenate([cols, v[:, 6:9]], 1)
    if multiview is not None:
        cols = np.concatenate([cols, multiview], 1)
    if use_height:
        floor = np.percentile(cols[:, 2], 0.99)
        cols = np.concatenate([cols, np.expand_dims(cols[:, 2] - floor, 1)], 1)
    return cols


class ResidentScan:
    """One scan with its point features in HBM; the label arrays are kept on the host (host-RNG mode) and, as slot ids
    + semantic ids, in HBM too (fully device-side mode)."""

    def __init__(self, raw, device, use_color=True, use_normal=False, use_height=True, multiview=None):
        pc = np.ascontiguousarray(point_features(raw["mesh_vertices"], use_color, use_normal, use_height, multiview))
        if pc.dtype not in (np.float32, np.float64):
            pc = pc.astype(np.float64)
        self.points = torch.from_numpy(pc).to(device)
        self.instance_labels = np.asarray(raw["instance_labels"])
        self.semantic_labels = np.asarray(raw["semantic_labels"])
        self.instance_bboxes = np.asarray(raw["instance_bboxes"])
        # for the fully device-side mode (build_batch_device): every distinct instance label of the scan is a slot
        # (ascending label = the order np.unique gives the reference); labels as slot ids + semantic ids in HBM
        self.slots, slot_of_vertex = np.unique(self.instance_labels, return_inverse=True)
        self.slot_of_vertex = torch.from_numpy(slot_of_vertex.astype(np.int32)).to(device)
        self.semantic_dev = torch.from_numpy(self.semantic_labels.astype(np.int32)).to(device)

    @property
    def num_vertices(self):
        return self.points.shape[0]


def _rotation(axis, t):
    """utils/pc_utils.py rotx / roty / rotz."""
    c, s = np.cos(t), np.sin(t)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def _rotate_boxes(boxes, rot, axis):
    """Axis-aligned boxes under a rotation about `axis` (model_util_scannet.py:51-83): centres rotate; the two
    extents across the axis become those of the rotated rectangle's bounding rectangle."""
    a, b = ((1, 2), (0, 2), (0, 1))[axis]
    half = np.stack([boxes[:, 3 + a], boxes[:, 3 + b]], 1) / 2.0
    ext = np.zeros((boxes.shape[0], 2, 4))
    for k, (sa, sb) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1))):
        corner = np.zeros((boxes.shape[0], 3))
        corner[:, 0], corner[:, 1] = sa * half[:, 0], sb * half[:, 1]
        corner = np.dot(corner, rot.T)
        ext[:, 0, k], ext[:, 1, k] = corner[:, 0], corner[:, 1]
    out = np.concatenate([np.dot(boxes[:, 0:3], rot.T), boxes[:, 3:6]], 1)
    out[:, 3 + a], out[:, 3 + b] = 2.0 * ext[:, 0].max(1), 2.0 * ext[:, 1].max(1)
    return out


def _augment_and_box_labels(d, scan, object_id, tables, augment):
    """lib/dataset.py:145-198 on the (<= 128) boxes: augmentation draws from torch's default generator in the
    reference's order (stored on `d` for the device kernel) and the box / reference-target labels."""
    boxes = scan.instance_bboxes
    nb = min(boxes.shape[0], MAX_NUM_OBJ)
    target = np.zeros((MAX_NUM_OBJ, 6))
    target[:nb] = boxes[:MAX_NUM_OBJ, 0:6]
    d.flip_x = d.flip_y = False
    d.rot, d.shift = [], None
    if augment:
        if torch.rand(1).item() > 0.5:
            d.flip_x = True
            target[:, 0] = -target[:, 0]
        if torch.rand(1).item() > 0.5:
            d.flip_y = True
            target[:, 1] = -target[:, 1]
        for axis in range(3):
            rot = _rotation(axis, (torch.rand(1).item() * np.pi / 18) - np.pi / 36)
            d.rot.append(rot)
            target = _rotate_boxes(target, rot, axis)
        d.shift = np.asarray((torch.rand(3) - 0.5).tolist(), np.float64)
        target[:, :3] += d.shift
    box_cls = [int(tables.nyu40id2class[int(x)]) for x in boxes[:nb, -2]]
    size_cls = np.zeros((MAX_NUM_OBJ,))
    size_res = np.zeros((MAX_NUM_OBJ, 3))
    size_cls[:nb] = box_cls
    size_res[:nb] = target[:nb, 3:6] - tables.mean_size_arr[box_cls]
    ref_box = np.zeros(MAX_NUM_OBJ)
    ref_center, ref_cls, ref_res = np.zeros(3), 0, np.zeros(3)
    for i, gt in enumerate(boxes[:nb, -1]):
        if gt == object_id:
            ref_box[i] = 1
            ref_center, ref_cls, ref_res = target[i, 0:3], size_cls[i], size_res[i]
    d.labels = dict(center_label=target.astype(np.float32)[:, 0:3], size_class_label=size_cls.astype(np.int64),
                    size_residual_label=size_res.astype(np.float32), num_bbox=np.array(nb).astype(np.int64),
                    ref_box_label=ref_box.astype(np.int64), ref_center_label=ref_center.astype(np.float32),
                    ref_size_class_label=np.array(int(ref_cls)).astype(np.int64),
                    ref_size_residual_label=ref_res.astype(np.float32),
                    ref_heading_class_label=np.array(0).astype(np.int64),
                    ref_heading_residual_label=np.array(0).astype(np.int64))



def _augment_and_box_labels_batch(draws, scans, object_ids, tables, augment):
    """The same labels for a whole batch with a handful of vectorised numpy calls (device mode: ~0.1 ms per SAMPLE of
    small numpy / torch calls otherwise). One torch.rand((B, 8)) replaces the per-sample draws, so the augmentation
    parameters are NOT on the reference's random stream (nothing in device mode is)."""
    B = len(scans)
    target = np.zeros((B, MAX_NUM_OBJ, 6))
    nbs = []
    for i, sc in enumerate(scans):
        nb = min(sc.instance_bboxes.shape[0], MAX_NUM_OBJ)
        target[i, :nb] = sc.instance_bboxes[:MAX_NUM_OBJ, 0:6]
        nbs.append(nb)
    rots = np.zeros((B, 3, 3, 3))
    shifts = None
    flips = np.zeros((B, 2), bool)
    if augment:
        r = torch.rand((B, 8)).numpy().astype(np.float64)
        flips = r[:, 0:2] > 0.5
        target[:, :, 0] *= np.where(flips[:, 0], -1.0, 1.0)[:, None]
        target[:, :, 1] *= np.where(flips[:, 1], -1.0, 1.0)[:, None]
        ang = r[:, 2:5] * np.pi / 18 - np.pi / 36
        c, s = np.cos(ang), np.sin(ang)
        one, zero = np.ones(B), np.zeros(B)
        rots[:, 0] = np.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1).reshape(B, 3, 3)
        rots[:, 1] = np.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1).reshape(B, 3, 3)
        rots[:, 2] = np.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1).reshape(B, 3, 3)
        sgn = np.array([(-1, -1), (1, -1), (1, 1), (-1, 1)], np.float64)             # corner signs
        m = max(nbs) if nbs else 0                                                     # rows beyond are zero boxes
        tv = target[:, :m]
        for axis, (a, b) in enumerate(((1, 2), (0, 2), (0, 1))):
            Rt = rots[:, axis].transpose(0, 2, 1)                                      # p' = p @ R.T, batched matmul
            half = np.stack([tv[:, :, 3 + a], tv[:, :, 3 + b]], 2) / 2.0              # (B, m, 2)
            corner = np.zeros((B, m, 4, 3))
            corner[..., 0] = half[:, :, None, 0] * sgn[None, None, :, 0]
            corner[..., 1] = half[:, :, None, 1] * sgn[None, None, :, 1]
            corner = np.matmul(corner.reshape(B, m * 4, 3), Rt).reshape(B, m, 4, 3)
            tv[:, :, 0:3] = np.matmul(tv[:, :, 0:3], Rt)
            tv[:, :, 3 + a] = 2.0 * corner[..., 0].max(2)
            tv[:, :, 3 + b] = 2.0 * corner[..., 1].max(2)
        shifts = r[:, 5:8].astype(np.float32).astype(np.float64) - 0.5
        target[:, :, :3] += shifts[:, None, :]
    for i, (d, sc, oid) in enumerate(zip(draws, scans, object_ids)):
        nb, boxes = nbs[i], sc.instance_bboxes
        d.flip_x, d.flip_y = bool(flips[i, 0]), bool(flips[i, 1])
        d.rot = [rots[i, 0], rots[i, 1], rots[i, 2]] if augment else []
        d.shift = shifts[i] if augment else None
        box_cls = tables.nyu40id2class[boxes[:nb, -2].astype(np.int64)]
        size_cls = np.zeros((MAX_NUM_OBJ,))
        size_res = np.zeros((MAX_NUM_OBJ, 3))
        size_cls[:nb] = box_cls
        size_res[:nb] = target[i, :nb, 3:6] - tables.mean_size_arr[box_cls]
        ref_box = np.zeros(MAX_NUM_OBJ)
        ref_center, ref_cls, ref_res = np.zeros(3), 0, np.zeros(3)
        hit = np.flatnonzero(boxes[:nb, -1] == oid)
        if len(hit):
            ref_box[hit] = 1
            j = hit[-1]                                  # the reference's loop keeps the last match
            ref_center, ref_cls, ref_res = target[i, j, 0:3], size_cls[j], size_res[j]
        d.labels = dict(center_label=target[i].astype(np.float32)[:, 0:3], size_class_label=size_cls.astype(np.int64),
                        size_residual_label=size_res.astype(np.float32), num_bbox=np.array(nb).astype(np.int64),
                        ref_box_label=ref_box.astype(np.int64), ref_center_label=ref_center.astype(np.float32),
                        ref_size_class_label=np.array(int(ref_cls)).astype(np.int64),
                        ref_size_residual_label=ref_res.astype(np.float32),
                        ref_heading_class_label=np.array(0).astype(np.int64),
                        ref_heading_residual_label=np.array(0).astype(np.int64))


class SampleDraw:
    """Host half of one sample: every random draw + the integer bookkeeping the device kernels need."""
    __slots__ = ("scan", "choices", "flip_x", "flip_y", "rot", "shift", "order", "seg", "rows", "classes", "labels",
                 "instance_labels")


def draw_sample(scan, object_id, tables, num_points=40000, augment=False):
    """Consumes the RNG streams exactly like lib/dataset.py:124 (scene choice), :154-181 (flips, three angles, shift)
    and :224 (one choice per object instance, ascending instance id)."""
    d = SampleDraw()
    d.scan = scan
    V = scan.num_vertices
    choices = np.random.choice(V, num_points, replace=V < num_points)
    d.choices = choices.astype(np.int32)
    ins = scan.instance_labels[choices]
    sem = scan.semantic_labels[choices]
    d.instance_labels = ins.astype(np.int64)

    _augment_and_box_labels(d, scan, object_id, tables, augment)

    # instance segments: stable sort by label == np.nonzero(labels == id) per ascending id (lib/dataset.py:207-210)
    small = ins.size and ins.min() >= -32768 and ins.max() <= 32767      # int16 keys take numpy's radix sort (6x)
    order = np.argsort(ins.astype(np.int16) if small else ins, kind="stable")
    sl = ins[order]
    starts = np.flatnonzero(np.concatenate(([True], sl[1:] != sl[:-1])))
    ends = np.concatenate((starts[1:], [len(sl)]))
    keep_order, seg, rows, classes = [], [0], [], []
    for lo, hi in zip(starts, ends):
        nyu = sem[order[lo]]
        if not tables.is_object(nyu):
            continue
        classes.append(int(tables.nyu40id2class[int(nyu)]))
        idx = order[lo:hi]
        keep_order.append(idx)
        seg.append(seg[-1] + (hi - lo))
        n_i = hi - lo
        rows.append(idx[np.random.choice(n_i, NUM_INSTANCE_POINTS, replace=n_i < NUM_INSTANCE_POINTS)])
    d.order = (np.concatenate(keep_order) if keep_order else np.zeros(0)).astype(np.int32)
    d.seg = np.asarray(seg, np.int32)
    d.rows = (np.stack(rows, 0) if rows else np.zeros((0, NUM_INSTANCE_POINTS))).astype(np.int32)
    d.classes = classes
    return d


class PendingBatch:
    """Everything of a batch enqueued; `finish()` waits for the (tiny) box / extent read-back and assembles the
    data_dict entries. Between build_batch() and finish() the host is free (e.g. to issue the previous step)."""

    def __init__(self, draws, clouds, inst_points, obbs_dev, extent_dev, host_back, event, voxel_size, device, keep):
        self.draws, self.clouds, self.inst_points = draws, clouds, inst_points
        self.obbs_dev, self.extent_dev, self.host_back, self.event = obbs_dev, extent_dev, host_back, event
        self.voxel_size, self.device, self._keep = voxel_size, device, keep

    def finish(self, data_dict=None):
        self.event.synchronize()
        B, S = len(self.draws), self.inst_points.shape[0]
        back = self.host_back.numpy()
        return self._assemble(data_dict, back[:S * 7].reshape(S, 7).copy(), back[S * 7:].reshape(B, 6).copy())

    def _assemble(self, data_dict, obbs, ext):
        from .sparse.utils import voxelize, voxelize_launch
        B = len(self.draws)
        dd = {} if data_dict is None else data_dict
        n, c = self.clouds.shape[1], self.clouds.shape[2]
        flat = self.clouds.view(B * n, c)
        batch = torch.arange(B, device=self.device, dtype=torch.int32).repeat_interleave(n)
        if _VOXELIZE_LAUNCH:
            # dev (IRX_INPUT_VOXELIZE_LAUNCH=1): the scene voxeliser without its host sync — a sparse.utils.VoxelizePending that
            # InstanceRefer.prepare_launch / prepare_finish turn into the canonical tensor when the level sizes have arrived
            dd["lidar"] = voxelize_launch(flat[:, :3].contiguous(), flat, batch, [self.voxel_size] * 3, B, 4)
        else:
            dd["lidar"] = voxelize(flat[:, :3].contiguous(), flat, batch, [self.voxel_size] * 3, B)
        classes, scene_of, start = [], [], [0]
        for i, d in enumerate(self.draws):
            classes += d.classes
            scene_of += [i] * len(d.classes)
            start.append(len(classes))
        dd["irx"] = InstancePack.from_device(self.inst_points, obbs, self.obbs_dev, classes, scene_of, start)
        dtype = self.clouds.dtype
        dd["point_min"] = self.extent_dev[:, :3].contiguous()
        dd["point_max"] = self.extent_dev[:, 3:].contiguous()
        np_dtype = np.float32 if dtype == torch.float32 else np.float64
        host = dict(point_min=ext[:, :3].astype(np_dtype), point_max=ext[:, 3:].astype(np_dtype))
        for k in ("ref_center_label", "ref_size_residual_label", "ref_size_class_label", "ref_heading_class_label",
                  "ref_heading_residual_label", "ref_box_label", "center_label", "size_class_label",
                  "size_residual_label", "num_bbox"):
            host[k] = np.stack([d.labels[k] for d in self.draws], 0)
        dd["_host"] = dict(dd.get("_host", {}), **host)
        for k in ("ref_center_label", "ref_size_residual_label"):
            dd[k] = idx_tensor(host[k], self.device, dtype=torch.from_numpy(host[k]).dtype)    # (pinned staging, as above)
        for k in ("ref_size_class_label", "ref_heading_class_label", "ref_heading_residual_label"):
            dd[k] = torch.from_numpy(host[k])
        dd["point_clouds"] = self.clouds
        dd["instance_labels"] = [d.instance_labels for d in self.draws]
        return dd


def build_batch(draws, device, voxel_size_glp=0.05):
    """Enqueue the device half for a list of SampleDraw (one per sample of the batch) on the current stream.
    -> PendingBatch. All samples must share num_points and the scans' feature width / dtype."""
    B = len(draws)
    assert B > 0
    pts0 = draws[0].scan.points
    c, dtype = pts0.shape[1], pts0.dtype
    eb = 4 if dtype == torch.float32 else 8
    n = len(draws[0].choices)
    for d in draws:
        assert len(d.choices) == n and d.scan.points.shape[1] == c and d.scan.points.dtype == dtype
    S = sum(len(d.classes) for d in draws)
    # ONE pinned staging buffer + ONE H2D copy for every index array of the batch
    sizes = [(len(d.choices), len(d.order), len(d.seg), d.rows.size) for d in draws]
    total = sum(sum(s) for s in sizes)
    stage = torch.empty(max(total, 1), dtype=torch.int32, pin_memory=True)
    sv = stage.numpy()
    o, offs = 0, []
    for d, (a, b, s, r) in zip(draws, sizes):
        sv[o:o + a] = d.choices
        sv[o + a:o + a + b] = d.order
        sv[o + a + b:o + a + b + s] = d.seg
        sv[o + a + b + s:o + a + b + s + r] = d.rows.reshape(-1)
        offs.append((o, o + a, o + a + b, o + a + b + s))
        o += a + b + s + r
    idx = stage.to(device, non_blocking=True)
    clouds = torch.empty((B, n, c), dtype=dtype, device=device)
    inst_points = torch.empty((S, NUM_INSTANCE_POINTS, c), dtype=dtype, device=device)
    obbs_dev = torch.empty((max(S, 1), 7), dtype=torch.float64, device=device)
    extent_dev = torch.empty((B, 6), dtype=dtype, device=device)
    stream = _lib.stream_ptr()
    base = idx.data_ptr()
    s0 = 0
    for i, (d, (oc, oo, os_, orow)) in enumerate(zip(draws, offs)):
        ni = len(d.classes)
        rot = np.ascontiguousarray(np.stack(d.rot, 0).reshape(-1)) if d.rot else None
        _lib.call("irx_scene_sample", _lib.ptr(d.scan.points), d.scan.num_vertices, c, base + 4 * oc, n,
                  int(d.flip_x), int(d.flip_y), rot.ctypes.data if rot is not None else None, len(d.rot),
                  d.shift.ctypes.data if d.shift is not None else None, clouds[i].data_ptr(), eb, stream)
        _lib.call("irx_instance_split", clouds[i].data_ptr(), n, c, base + 4 * oo, base + 4 * os_, ni,
                  base + 4 * orow, NUM_INSTANCE_POINTS,
                  inst_points[s0:].data_ptr() if ni else None, obbs_dev[s0:].data_ptr(), extent_dev[i].data_ptr(),
                  eb, stream)
        s0 += ni
    # boxes + extents back to the host in one async copy
    back_dev = torch.cat([obbs_dev[:S].reshape(-1), extent_dev.double().reshape(-1)])
    host_back = torch.empty(back_dev.shape, dtype=torch.float64, pin_memory=True)
    host_back.copy_(back_dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingBatch(draws, clouds, inst_points, obbs_dev[:S], extent_dev, host_back, ev, voxel_size_glp, device,
                        (idx, back_dev, stage))


# ---- fully device-side mode: no per-point work on the host at all ----------------------------------------------------
class _Draw:
    """The host part that remains in device mode: augmentation draws + box labels (<= 128 boxes)."""
    __slots__ = ("scan", "flip_x", "flip_y", "rot", "shift", "labels", "classes", "instance_labels")


class PendingDeviceBatch:
    def __init__(self, draws, clouds, inst_points, obbs_dev, extent_dev, host_back, event, slot_base, voxel_size, device,
                 keep):
        self.draws, self.clouds, self.inst_points, self.obbs_dev = draws, clouds, inst_points, obbs_dev
        self.extent_dev, self.host_back, self.event, self.slot_base = extent_dev, host_back, event, slot_base
        self.voxel_size, self.device, self._keep = voxel_size, device, keep

    def finish(self, data_dict=None):
        """Collect (count, class) per slot + boxes + extents (one async copy, already under way), drop the slots that
        are not object instances of this sample, assemble the data_dict entries."""
        self.event.synchronize()
        B, S = len(self.draws), self.inst_points.shape[0]
        back = self.host_back.numpy()
        counts, cls = back[:S].astype(np.int64), back[S:2 * S].astype(np.int64)
        obbs = back[2 * S:2 * S + 7 * S].reshape(S, 7)
        ext = back[9 * S:].reshape(B, 6).copy()
        keep = np.flatnonzero((counts > 0) & (cls >= 0))
        kept_dev = idx_tensor(keep, self.device)       # (pinned staging: a pageable H2D copy blocks until the stream drains)
        inst_points = self.inst_points.index_select(0, kept_dev)
        obbs_dev = self.obbs_dev.index_select(0, kept_dev)
        for i, d in enumerate(self.draws):
            mine = keep[(keep >= self.slot_base[i]) & (keep < self.slot_base[i + 1])]
            d.classes = [int(c) for c in cls[mine]]
            d.instance_labels = None
        shim = PendingBatch(self.draws, self.clouds, inst_points, obbs_dev, self.extent_dev, None, None, self.voxel_size,
                            self.device, None)
        return shim._assemble(data_dict, obbs[keep].copy(), ext)


def build_batch_device(scans, object_ids, tables, device, num_points=40000, augment=False, voxel_size_glp=0.05,
                       seed=None):
    """SURVEY §8(f) rank 1 as written: everything per-point on the GPU from the resident scan + its label arrays.
    Scene sub-sampling and the 1024-point resample use counter-based draws on the device (irx_random_subset /
    irx_resample_rows: keyed pseudo-random permutations, no sort, no generator state): the same distribution family as
    the reference (a subset without replacement; per instance 1024 distinct rows, or 1024 draws with replacement when
    it has fewer points) but not numpy's random stream — use draw_sample() + build_batch() when a run must reproduce
    the reference sample for sample. `seed`: int for a reproducible batch, None = one draw from torch's CPU generator.
    Host work per sample: the augmentation draws and the (<= 128) box labels only; no host sync anywhere.
    -> PendingDeviceBatch (finish() as in build_batch)."""
    B = len(scans)
    assert 0 < B <= 64, "build_batch_device: 1..64 samples per call (irx_random_subset)"
    pts0 = scans[0].points
    c, dtype = pts0.shape[1], pts0.dtype
    for sc in scans:
        assert sc.points.shape[1] == c and sc.points.dtype == dtype, "scans of a batch share feature width and dtype"
    eb = 4 if dtype == torch.float32 else 8
    n = num_points
    draws = []
    for sc in scans:
        d = _Draw()
        d.scan = sc
        draws.append(d)
    _augment_and_box_labels_batch(draws, scans, object_ids, tables, augment)
    stream = _lib.stream_ptr()
    clouds = torch.empty((B, n, c), dtype=dtype, device=device)
    slot_base = np.concatenate([[0], np.cumsum([len(sc.slots) for sc in scans])]).astype(np.int64)
    S = int(slot_base[-1])
    gslot = torch.empty((B, n), dtype=torch.int64, device=device)       # global slot id of every sampled point
    sem = torch.empty((B, n), dtype=torch.int64, device=device)
    import ctypes
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # CPU tensor: no device sync
    seeds = [(seed * 0x9E3779B97F4A7C15 + 0xBF58476D1CE4E5B9 * (i + 1)) & 0xFFFFFFFFFFFFFFFF for i in range(B + 1)]
    choices_all = torch.empty((B, n), dtype=torch.int64, device=device)
    _lib.call("irx_random_subset", B, (ctypes.c_int * B)(*[sc.num_vertices for sc in scans]), n,
              (ctypes.c_uint64 * B)(*seeds[:B]), choices_all.data_ptr(), stream)
    choices = [choices_all[i] for i in range(B)]
    P, I, D, L = ctypes.c_void_p * B, ctypes.c_int * B, ctypes.c_double, ctypes.c_int64 * B
    rots = np.zeros((B, 27))
    shifts = np.zeros((B, 3))
    for i, d in enumerate(draws):
        if d.rot:
            rots[i, :9 * len(d.rot)] = np.stack(d.rot, 0).reshape(-1)
        if d.shift is not None:
            shifts[i] = d.shift
    flips = (ctypes.c_int * (2 * B))(*[int(v) for d in draws for v in (d.flip_x, d.flip_y)])
    _lib.call("irx_scene_sample_batch", B, P(*[sc.points.data_ptr() for sc in scans]),
              I(*[sc.num_vertices for sc in scans]), c, P(*[ch.data_ptr() for ch in choices]), n, flips,
              rots.ctypes.data, I(*[len(d.rot) for d in draws]), shifts.ctypes.data,
              I(*[int(d.shift is not None) for d in draws]), clouds.data_ptr(),
              P(*[sc.slot_of_vertex.data_ptr() for sc in scans]), P(*[sc.semantic_dev.data_ptr() for sc in scans]),
              L(*[int(v) for v in slot_base[:-1]]), gslot.data_ptr(), sem.data_ptr(), eb, stream)
    gs = gslot.view(-1)
    # points grouped by slot: a stable sort on 16-bit keys (ascending point index inside a slot, like np.nonzero); the
    # segment bounds come from a binary search in the sorted keys and the first point of a slot is the segment's head
    # (bincount / scatter-amin would be 800 k atomics on ~150 addresses: measured 1.7 ms)
    from .sparse import functional as F_
    skeys, order32 = F_.sort_keys(gs.to(torch.int64), max(int(S).bit_length(), 1))   # the build's stable radix sort
    order = order32.long()
    seg = torch.searchsorted(skeys, torch.arange(S + 1, device=device, dtype=torch.int64))
    seg32 = seg.to(torch.int32)
    counts = seg[1:] - seg[:-1]
    # semantic id of each instance's FIRST sampled point (lib/dataset.py:213: semantic_labels[ind[0]])
    first = order.index_select(0, seg[:-1].clamp(max=B * n - 1))
    nyu = sem.view(-1).index_select(0, first)
    lut = tables.__dict__.get("_lut_dev")
    if lut is None or lut.device != torch.device(device):
        # once per ClassTables and device: a pageable H2D copy blocks the issuing thread until its stream has drained (4 ms per
        # batch of the preparation worker's time in the round-5 host profile of tools/e2e_train_bench.py)
        lut = torch.from_numpy(np.where(tables._is_object[:len(tables.nyu40id2class)], tables.nyu40id2class, -1)
                               .astype(np.int64)).to(device)
        tables.__dict__["_lut_dev"] = lut
    cls = torch.where((nyu >= 0) & (nyu < lut.shape[0]), lut[nyu.clamp(0, lut.shape[0] - 1)], torch.full_like(nyu, -1))
    rows32 = torch.empty((S, NUM_INSTANCE_POINTS), dtype=torch.int32, device=device)
    _lib.call("irx_resample_rows", _lib.ptr(order32), _lib.ptr(seg32), S, NUM_INSTANCE_POINTS,
              ctypes.c_uint64(seeds[B]), _lib.ptr(rows32), stream)
    inst_points = torch.empty((S, NUM_INSTANCE_POINTS, c), dtype=dtype, device=device)
    obbs_dev = torch.empty((max(S, 1), 7), dtype=torch.float64, device=device)
    _lib.call("irx_instance_split", clouds.data_ptr(), B * n, c, _lib.ptr(order32), _lib.ptr(seg32), S,
              _lib.ptr(rows32), NUM_INSTANCE_POINTS, inst_points.data_ptr() if S else None, obbs_dev.data_ptr(), None,
              eb, stream)
    xyz = clouds[:, :, :3]
    extent = torch.cat([xyz.amin(1), xyz.amax(1)], 1)
    back_dev = torch.cat([counts.double(), cls.double(), obbs_dev[:S].reshape(-1), extent.double().reshape(-1)])
    host_back = torch.empty(back_dev.shape, dtype=torch.float64, pin_memory=True)
    host_back.copy_(back_dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingDeviceBatch(draws, clouds, inst_points, obbs_dev[:S], extent, host_back, ev, slot_base, voxel_size_glp, device,
                              (back_dev, rows32, order32, seg32, gslot, sem, choices_all))


MAX_DES_LEN = 126                                   # lib/config.py CONF.TRAIN.MAX_DES_LEN


def embed_tokens(tokens, glove, max_len=MAX_DES_LEN, dim=300):
    """Token list -> (`lang_feat` (max_len, dim) float32, `lang_len`): the language half of the reference's __getitem__
    (lib/dataset.py:70-92,275-277; the same loop again in `_tranform_des` :403-413). Row t holds glove[token_t], glove["unk"] for
    an out-of-vocabulary token, zeros for a whitespace token (its row is skipped, not removed) and beyond the utterance;
    `lang_len` counts the non-whitespace tokens of the WHOLE utterance, capped at max_len. Host code: a dictionary lookup per
    token, done once per sample when the sample list of ResidentLoader is built."""
    feat = np.zeros((max_len, dim), np.float64)
    for t, tok in enumerate(tokens[:max_len]):
        if tok.isspace():
            continue
        feat[t] = glove[tok] if tok in glove else glove["unk"]
    n = sum(1 for tok in tokens if not tok.isspace())
    return feat.astype(np.float32), np.int64(min(n, max_len))


class ResidentLoader:
    """Iterable of training batches built on the device from resident scans (build_batch_device), in the role of the
    reference's DataLoader(ScannetReferenceDataset, collate_fn) for lib/solver.py-style loops (instancerefer_amd.solver.
    Solver accepts it as dataloader["train"]). `samples`: list of dicts with `scan` (key into `scans`), `object_id`,
    `object_cat`, `lang_feat` (126, 300) float32, `lang_len`, optional `unique_multiple`. The input pipeline of batch
    i + 1 is enqueued before batch i is handed out, so its kernels overlap the training step; no host sync besides the
    small read-back collected in finish()."""

    def __init__(self, scans, samples, tables, batch_size, device, num_points=40000, augment=False, shuffle=True,
                 seed=0, drop_last=True, voxel_size_glp=0.05):
        self.scans, self.samples, self.tables, self.batch_size, self.device = scans, samples, tables, batch_size, device
        self.num_points, self.augment, self.shuffle, self.seed, self.drop_last = num_points, augment, shuffle, seed, drop_last
        self.voxel_size_glp = voxel_size_glp
        self.epoch = 0

    def __len__(self):
        n = len(self.samples)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _launch(self, idx, seed):
        batch = [self.samples[i] for i in idx]
        pend = build_batch_device([self.scans[b["scan"]] for b in batch], [b["object_id"] for b in batch], self.tables,
                                  self.device, self.num_points, self.augment, self.voxel_size_glp, seed=seed)
        return pend, batch

    def _finish(self, pend, batch):
        dd = pend.finish()
        B = len(batch)
        lang = np.stack([np.asarray(b["lang_feat"], np.float32) for b in batch], 0)
        lens = np.asarray([int(b["lang_len"]) for b in batch], np.int64)
        cats = np.asarray([int(b["object_cat"]) for b in batch], np.int64)
        dd["lang_feat"] = torch.from_numpy(lang).to(self.device, non_blocking=True)
        dd["lang_len"] = torch.from_numpy(lens).to(self.device, non_blocking=True)
        dd["lang_len_max"] = int(lens.max())
        dd["object_cat"] = torch.from_numpy(cats).to(self.device, non_blocking=True)
        dd["unique_multiple"] = torch.from_numpy(np.asarray([int(b.get("unique_multiple", 0)) for b in batch], np.int64))
        dd["_host"].update(object_cat=cats, unique_multiple=dd["unique_multiple"].numpy())
        assert dd["point_clouds"].shape[0] == B
        return dd

    def __iter__(self):
        order = np.arange(len(self.samples))
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(order)
        self.epoch += 1
        chunks = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        base = (self.seed * 1000003 + self.epoch) * 4099
        pending = self._launch(chunks[0], base) if chunks else None
        for ci in range(len(chunks)):
            cur = pending
            pending = self._launch(chunks[ci + 1], base + ci + 1) if ci + 1 < len(chunks) else None
            yield self._finish(*cur)
