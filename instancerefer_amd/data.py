"""Host-side helpers around the reference's `data_dict` contract.

`upload_instances` moves the ragged python lists of per-instance (1024, C0) float64 point arrays that
lib/dataset.py:201-298 (reference) produces to the GPU ONCE per batch, so that candidate filtering, both
voxelisations and the per-instance means run on device. The drop-in modules accept a plain reference
data_dict too (they call this themselves when the `irx` entry is absent)."""
import numpy as np
import torch


class InstancePack:
    """All instances of a batch, flattened scene-major.
      xyz64   (S, P, 3)  float64 cuda — exact voxel assignment needs float64 (np.floor(x / v) parity)
      pts32   (S, P, C0) float32 cuda — features (first-point voxel features, per-instance means)
      obbs    (S, 7)     float64 numpy (host; returned to callers as pred_obb_batch)
      centres (S, 3)     float32 cuda
      classes list[int] length S ; scene_of list[int] length S ; scene_start list[int] length B+1
    """

    def __init__(self, data_dict, device):
        ipts = data_dict['instance_points']
        iobb = data_dict['instance_obbs']
        icls = data_dict['instance_class']
        self.batch_size = len(ipts)
        flat, obbs, classes, scene_of, start = [], [], [], [], [0]
        for i in range(self.batch_size):
            for j in range(len(ipts[i])):
                flat.append(np.asarray(ipts[i][j]))
                obbs.append(np.asarray(iobb[i][j], dtype=np.float64))
                classes.append(int(icls[i][j]))
                scene_of.append(i)
            start.append(len(flat))
        self.classes = classes
        self.scene_of = scene_of
        self.scene_start = start
        self.obbs = np.stack(obbs, 0) if obbs else np.zeros((0, 7))
        if flat:
            pts = torch.from_numpy(np.stack(flat, 0))            # (S, P, C0) float64, one H2D copy
            pts = pts.to(device, non_blocking=True)
            self.xyz64 = pts[:, :, :3].double().contiguous()
            self.pts32 = pts.float().contiguous()
        else:
            self.xyz64 = torch.zeros((0, 1, 3), dtype=torch.float64, device=device)
            self.pts32 = torch.zeros((0, 1, 3), dtype=torch.float32, device=device)
        self.centres = torch.from_numpy(self.obbs[:, :3].astype(np.float32)).to(device)
        # (S, 7) float64 boxes for the device-side IoU labelling / evaluation (irx_iou_labels, irx_eval_select)
        self.obbs_dev = torch.from_numpy(np.ascontiguousarray(self.obbs)).to(device, non_blocking=True) if flat else \
            torch.zeros((0, 7), dtype=torch.float64, device=device)
        self._sel_cache = {}

    @classmethod
    def from_device(cls, inst_points, obbs_host, obbs_dev, classes, scene_of, scene_start):
        """Pack built by the device-side input pipeline (scene_input.build_batch): the instance points never existed
        on the host. inst_points (S, P, C0) cuda f32/f64; obbs_host (S, 7) numpy; obbs_dev (S, 7) cuda f64."""
        self = cls.__new__(cls)
        self.batch_size = len(scene_start) - 1
        self.classes, self.scene_of, self.scene_start = list(classes), list(scene_of), list(scene_start)
        self.obbs = np.asarray(obbs_host, np.float64).reshape(-1, 7)
        self.obbs_dev = obbs_dev.double().contiguous()
        if inst_points.shape[0]:
            self.xyz64 = inst_points[:, :, :3].double().contiguous()
            self.pts32 = inst_points.float().contiguous()
            self.centres = obbs_dev[:, :3].float().contiguous()
        else:
            dev = inst_points.device
            self.xyz64 = torch.zeros((0, 1, 3), dtype=torch.float64, device=dev)
            self.pts32 = torch.zeros((0, 1, 3), dtype=torch.float32, device=dev)
            self.centres = torch.zeros((0, 3), dtype=torch.float32, device=dev)
        self._sel_cache = {}
        return self

    def select(self, lang_cls_pred):
        """Candidate filtering of AttributeModule.filter_candidates / RelationModule.filter_candidates
        (reference attribute_module.py:42-81, relation_module.py:38-78) on host integers only.
        -> dict with
             cand        flat indices of same-class instances of scenes with >= 2 candidates (batch-major)
             cand_scene  scene index per candidate
             pred_obb_batch list[B] of (c_i, 7) float64 arrays ((0,) when c_i == 0)
             num_filtered_objs list[B]
             support     flat indices of ALL instances of scenes with >= 2 candidates
             support_scene_offsets int list (len = #kept scenes + 1), support_batch list (scene idx)
             query_in_support index of every candidate inside `support`
        """
        key = tuple(int(v) for v in lang_cls_pred)
        if key in self._sel_cache:
            return self._sel_cache[key]
        cand, cand_scene, pred_obb_batch, nfo, filtered = [], [], [], [], []
        support, support_batch, query_in_support, sup_off = [], [], [], [0]
        for i in range(self.batch_size):
            lo, hi = self.scene_start[i], self.scene_start[i + 1]
            mine = [s for s in range(lo, hi) if self.classes[s] == key[i]]
            nfo.append(len(mine))
            filtered += mine
            pred_obb_batch.append(np.asarray([self.obbs[s] for s in mine]))
            if len(mine) < 2:
                continue
            cand += mine
            cand_scene += [i] * len(mine)
            base = len(support)
            support += list(range(lo, hi))
            support_batch += [i] * (hi - lo)
            query_in_support += [base + (s - lo) for s in mine]
            sup_off.append(len(support))
        sel = dict(cand=cand, cand_scene=cand_scene, pred_obb_batch=pred_obb_batch, num_filtered_objs=nfo,
                   support=support, support_batch=support_batch, query_in_support=query_in_support,
                   support_scene_offsets=sup_off, filtered=filtered)
        self._sel_cache[key] = sel
        return sel


def upload_instances(data_dict, device=None):
    """Attach the device-resident instance pack as data_dict['irx'] (idempotent)."""
    if 'irx' not in data_dict:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        data_dict['irx'] = InstancePack(data_dict, device)
    return data_dict['irx']


def idx_tensor(values, device, dtype=torch.int64):
    """Small host list / array -> device tensor WITHOUT stalling the host: a pageable H2D copy blocks until the stream
    has drained (measured ~0.3 ms each behind a full queue), so stage through the caching pinned allocator and copy
    asynchronously."""
    device = torch.device(device)
    src = torch.as_tensor(np.asarray(values), dtype=dtype)
    if device.type != "cuda":
        return src.to(device)
    pinned = torch.empty(src.shape, dtype=dtype, pin_memory=True)
    pinned.copy_(src)
    return pinned.to(device, non_blocking=True)


def pack_to_device(arrays, device):
    """Several small host integer arrays -> device int64 tensors with ONE pinned staging buffer and ONE async H2D copy
    (each separate upload costs ~25 us of host time). Returns (list of 1-D int64 views, the backing device buffer)."""
    arrs = [np.asarray(a, dtype=np.int64).reshape(-1) for a in arrays]
    total = sum(a.size for a in arrs)
    device = torch.device(device)
    host = torch.empty(max(total, 1), dtype=torch.int64, pin_memory=(device.type == "cuda"))
    hv = host.numpy()
    offs, o = [], 0
    for a in arrs:
        hv[o:o + a.size] = a
        offs.append((o, a.size))
        o += a.size
    buf = host.to(device, non_blocking=True)
    return [buf[o:o + n] for o, n in offs], buf


def selection_on_device(sel, pack, device):
    """Device index tensors of a candidate selection (InstancePack.select), built once per selection:
    cand, cand_scene, support, support_class, support_seg (scene id renumbered over the kept scenes), query_in_support
    (int64) and support_offsets (int32). The training loop's input-preparation stage calls this (via
    AttributeModule.prepare) so that the uploads are off the training thread; the modules build it on demand otherwise."""
    dev = sel.get('_dev')
    if dev is not None and dev['buf'].device == torch.device(device):
        return dev
    kept = sel['support_scene_offsets']
    seg_of = np.repeat(np.arange(len(kept) - 1), np.diff(kept)) if len(kept) > 1 else np.zeros(0, np.int64)
    cls = [pack.classes[s] for s in sel['support']]
    views, buf = pack_to_device([sel['cand'], sel['cand_scene'], sel['support'], cls, seg_of, sel['query_in_support'], kept],
                                device)
    dev = dict(cand=views[0], cand_scene=views[1], support=views[2], support_class=views[3], support_seg=views[4],
               query_in_support=views[5], support_offsets=views[6].to(torch.int32), buf=buf)
    sel['_dev'] = dev
    return dev

