"""get_loss — drop-in for the reference's lib/loss_helper.py:196-269 (+ ContrastiveLoss :93-107,
compute_scene_mask_loss :131-161, compute_lang_classification_loss :189-193).

loss = 10 * ref_loss + lang_loss + seg_loss.  ref_loss: per sample with >= 2 candidates and max IoU >= 0.2,
ContrastiveLoss(margin .2, gamma 5) between the summed scores and the one-hot of the candidate with the
highest IoU against the GT box, divided by the FULL batch size. IoU labelling (axis-aligned IoU of box corners,
utils/box_util.py:154-175,310-333, float64): ONE device launch over all candidates of the batch (irx_iou_labels,
bit-identical to numpy) when the instance boxes are resident, float64 numpy on the host otherwise (CPU tensors).
`config` is duck-typed: anything with `param2obb_batch` (the reference's ScannetDatasetConfig works).
"""
import numpy as np
import torch
import torch.nn as nn


class DatasetConfig:
    """Minimal stand-in for data/scannet/model_util_scannet.py:85-181 (ScannetDatasetConfig): only what
    get_loss / get_eval use. mean_size_arr (num_class, 3); ScanNet boxes are axis-aligned (heading 0)."""

    def __init__(self, mean_size_arr=None, num_class=18):
        self.num_class = num_class
        self.num_heading_bin = 1
        self.num_size_cluster = num_class
        self.mean_size_arr = np.ones((num_class, 3)) if mean_size_arr is None else np.asarray(mean_size_arr)

    def class2angle_batch(self, pred_cls, residual, to_label_format=True):
        return np.zeros(pred_cls.shape[0])

    def class2size_batch(self, pred_cls, residual):
        return self.mean_size_arr[pred_cls] + residual

    def param2obb_batch(self, center, heading_class, heading_residual, size_class, size_residual):
        heading_angle = self.class2angle_batch(heading_class, heading_residual)
        obb = np.zeros((heading_class.shape[0], 7))
        obb[:, 0:3] = center
        obb[:, 3:6] = self.class2size_batch(size_class, size_residual)
        obb[:, 6] = heading_angle * -1
        return obb


def _roty_batch(t):
    out = np.zeros(tuple(list(t.shape) + [3, 3]))
    c, s = np.cos(t), np.sin(t)
    out[..., 0, 0] = c
    out[..., 0, 2] = s
    out[..., 1, 1] = 1
    out[..., 2, 0] = -s
    out[..., 2, 2] = c
    return out


def get_3d_box_batch(box_size, heading_angle, center):
    """(…,3),(…),(…,3) -> (…,8,3) corners (same corner order / rotation as the reference's box_util)."""
    shape = heading_angle.shape
    R = _roty_batch(heading_angle)
    l = np.expand_dims(box_size[..., 0], -1)
    w = np.expand_dims(box_size[..., 1], -1)
    h = np.expand_dims(box_size[..., 2], -1)
    corners = np.zeros(tuple(list(shape) + [8, 3]))
    corners[..., :, 0] = np.concatenate((l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2), -1)
    corners[..., :, 1] = np.concatenate((w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2), -1)
    corners[..., :, 2] = np.concatenate((h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2), -1)
    t = list(range(len(shape))) + [len(shape) + 1, len(shape)]
    corners = np.matmul(corners, np.transpose(R, tuple(t)))
    corners += np.expand_dims(center, -2)
    return corners


def box3d_iou_batch(corners1, corners2):
    """Axis-aligned IoU from (N,8,3) corners."""
    mn1, mx1 = corners1.min(axis=1), corners1.max(axis=1)
    mn2, mx2 = corners2.min(axis=1), corners2.max(axis=1)
    lo = np.maximum(mn1, mn2)
    hi = np.minimum(mx1, mx2)
    inter = np.maximum(hi[:, 0] - lo[:, 0], 0) * np.maximum(hi[:, 1] - lo[:, 1], 0) * np.maximum(hi[:, 2] - lo[:, 2], 0)
    v1 = (mx1[:, 0] - mn1[:, 0]) * (mx1[:, 1] - mn1[:, 1]) * (mx1[:, 2] - mn1[:, 2])
    v2 = (mx2[:, 0] - mn2[:, 0]) * (mx2[:, 1] - mn2[:, 1]) * (mx2[:, 2] - mn2[:, 2])
    return inter / (v1 + v2 - inter + 1e-8)


class ContrastiveLoss(nn.Module):
    """clamp(logsumexp(gamma*score * (1-label)) - sum(gamma*score*label) + margin, 0): the positive slot
    enters the log-sum-exp as exp(0), not -inf — reference quirk kept (loss_helper.py:101-106). Unlike the
    reference this does not scale `score` in place."""

    def __init__(self, margin=0.2, gamma=5, reduction='mean'):
        super().__init__()
        self.margin = margin
        self.gamma = gamma
        self.reduction = reduction

    def forward(self, score, label):
        score = score * self.gamma
        sim = (score * label).sum()
        neg_sim = torch.logsumexp(score * label.logical_not(), dim=0)
        return torch.clamp(neg_sim - sim + self.margin, min=0).sum()


def compute_scene_mask_loss(data_dict):
    """9-area label of the GT centre on the 3x3 xy grid of the scene's bounding box + CE."""
    pred = data_dict['seg_scores']
    dev = pred.device
    c = data_dict["ref_center_label"].to(dev)
    point_min = data_dict['point_min'].to(dev)
    point_max = data_dict['point_max'].to(dev)
    first = point_min + (point_max - point_min) / 3
    second = point_min + (point_max - point_min) / 3 * 2
    rf = torch.le(c, first)
    rs = torch.le(c, second)
    # column bin bx: 0 if x<=first, 1 if first<x<=second, 2 otherwise (same for y); label table as reference
    bx = (~rf[:, 0]).long() + (~rs[:, 0]).long()
    by = (~rf[:, 1]).long() + (~rs[:, 1]).long()
    label = by * 3 + bx
    loss = nn.functional.cross_entropy(pred, label)
    acc = (torch.argmax(pred, 1) == label).sum() / float(label.numel())
    return loss, acc


def compute_lang_classification_loss(data_dict):
    return nn.functional.cross_entropy(data_dict["lang_scores"], data_dict["object_cat"].to(data_dict["lang_scores"].device))


def _host_np(data_dict, key):
    """Label tensors originate on the host (dataloader); reuse a host copy when the caller kept one
    (synthetic.to_device / a Solver that stashes `_host`), else fall back to the reference's D2H copy."""
    h = data_dict.get("_host")
    if h is not None and key in h:
        return h[key]
    return data_dict[key].detach().cpu().numpy()


def _selection(data_dict):
    """The candidate selection of this batch (data.InstancePack.select) if the instance pack is attached."""
    prep = data_dict.get('_attr_prepared')
    if prep is not None:
        return prep[1]
    pack, cls = data_dict.get('irx'), data_dict.get('_lang_cls_pred_list')
    if pack is not None and cls is not None:
        return pack.select(cls)
    return None


def _prepare_labels_device(data_dict, out, ref_gt_obb, area_label, sel, pack, device):
    """Device half: counts / offsets are host integers of the selection; the IoUs, the arg-max labels and the IoU >= 0.2
    gate are ONE launch (csrc/irx_labels.hip) over boxes that are already resident (InstancePack.obbs_dev). One pinned
    staging buffer, one H2D copy (indices + the B ground-truth boxes as raw float64 bits), no D2H."""
    from . import _lib
    counts, batch_size, total = out['counts'], out['batch_size'], out['total']
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    scored_pos = np.full(batch_size, -1, np.int64)
    scored_row = np.full(batch_size, -1, np.int64)
    pos = row = 0
    for i, c in enumerate(counts):
        if c >= 2:
            scored_pos[i], scored_row[i] = pos, row
            pos += c
            row += 1
    nlab, srow = pos, row
    seg_off = np.concatenate([[0], np.cumsum([c for c in counts if c >= 2])]).astype(np.int64)
    nal = 0 if area_label is None else area_label.size
    parts = [np.asarray(sel['filtered'], np.int64), starts, scored_pos, scored_row,
             np.zeros(0, np.int64) if area_label is None else area_label.astype(np.int64), seg_off,
             np.ascontiguousarray(ref_gt_obb, dtype=np.float64).reshape(-1).view(np.int64)]
    ih = torch.empty(sum(a.size for a in parts), dtype=torch.int64, pin_memory=True)
    hv, offs, o = ih.numpy(), [], 0
    for a in parts:
        hv[o:o + a.size] = a
        offs.append((o, a.size))
        o += a.size
    buf = ih.to(device, non_blocking=True)
    v = [buf[a:a + n] for a, n in offs]
    gt_dev = v[6].view(torch.float64)
    fbuf = torch.empty(total + nlab + srow, dtype=torch.float32, device=device)
    labels, lab, keep = fbuf[:total], fbuf[total:total + nlab], fbuf[total + nlab:]
    _lib.call("irx_iou_labels", _lib.ptr(pack.obbs_dev), _lib.ptr(v[0]), _lib.ptr(v[1]), _lib.ptr(gt_dev), batch_size,
              _lib.ptr(v[2]) if srow else None, _lib.ptr(v[3]) if srow else None, _lib.ptr(labels),
              _lib.ptr(lab) if srow else None, _lib.ptr(keep) if srow else None, None, _lib.stream_ptr())
    out.update(starts=starts, srow=srow, lmax=0, fbuf=fbuf, buf=buf, label_dev=labels, lab=lab, keep_dev=keep, flat=None,
               area_label=v[4] if nal else None, seg_off=v[5],
               dev=dict(filtered=v[0], starts=v[1], scored_pos=v[2], gt=gt_dev, obbs=pack.obbs_dev))
    return out


def prepare_labels(data_dict, config, device=None):
    """Host half of get_loss: IoU labelling of every candidate box against the GT box (float64 numpy, as the reference)
    and ONE staged upload each of the integer / float label arrays. Depends only on the batch's inputs (GT labels +
    the candidate boxes chosen by the class filter), so a training loop can run it in its input-preparation stage
    (data_dict['_loss_prepared'] = prepare_labels(...)); get_loss does it itself otherwise."""
    if 'pred_obb_batch' in data_dict:
        pred_obb_batch = data_dict['pred_obb_batch']
    else:
        pred_obb_batch = data_dict['_attr_prepared'][1]['pred_obb_batch']
    if device is None:
        device = data_dict['point_min'].device if torch.is_tensor(data_dict.get('point_min')) else torch.device('cpu')
    ref_gt_obb = config.param2obb_batch(_host_np(data_dict, "ref_center_label"),
                                        _host_np(data_dict, "ref_heading_class_label"),
                                        _host_np(data_dict, "ref_heading_residual_label"),
                                        _host_np(data_dict, "ref_size_class_label"),
                                        _host_np(data_dict, "ref_size_residual_label"))
    ref_gt_bbox = get_3d_box_batch(ref_gt_obb[:, 3:6], ref_gt_obb[:, 6], ref_gt_obb[:, 0:3])   # (B, 8, 3)
    batch_size = len(pred_obb_batch)
    counts = [int(p.shape[0]) for p in pred_obb_batch]
    total = sum(counts)
    out = dict(batch_size=batch_size, counts=counts, total=total, srow=0, lmax=0, buf=None, fbuf=None, area_label=None)
    host = data_dict.get("_host") or {}
    if "point_min" in host and "point_max" in host:
        # 9-area label of the GT centre (compute_scene_mask_loss) from the host copies: same float64 comparisons
        c = np.asarray(_host_np(data_dict, "ref_center_label"), dtype=np.float64)
        pmin, pmax = np.asarray(host["point_min"], np.float64), np.asarray(host["point_max"], np.float64)
        first = pmin + (pmax - pmin) / 3
        second = pmin + (pmax - pmin) / 3 * 2
        bx = (c[:, 0] > first[:, 0]).astype(np.int64) + (c[:, 0] > second[:, 0]).astype(np.int64)
        by = (c[:, 1] > first[:, 1]).astype(np.int64) + (c[:, 1] > second[:, 1]).astype(np.int64)
        out["area_label_host"] = by * 3 + bx
    if total == 0:
        out.pop("area_label_host", None)
        return out
    sel, pack = _selection(data_dict), data_dict.get('irx')
    if (torch.device(device).type == "cuda" and sel is not None and 'filtered' in sel
            and getattr(pack, 'obbs_dev', None) is not None and pack.obbs_dev.is_cuda
            and list(sel['num_filtered_objs']) == counts
            # k_iou_labels / k_eval_select work on axis-aligned boxes (ScanRefer: every heading is 0, lib/dataset.py:216);
            # a rotated GT or candidate box takes the host path, which rotates the corners like get_3d_box_batch
            and not np.any(ref_gt_obb[:, 6]) and not any(np.any(p.reshape(-1, 7)[:, 6]) for p in pred_obb_batch if p.shape[0])):
        return _prepare_labels_device(data_dict, out, ref_gt_obb, out.pop("area_label_host", None), sel, pack,
                                      torch.device(device))
    obbs = np.concatenate([p.reshape(-1, 7) for p in pred_obb_batch if p.shape[0]], 0)       # (total, 7)
    scene_of = np.repeat(np.arange(batch_size), counts)
    pred_bbox = get_3d_box_batch(obbs[:, 3:6], obbs[:, 6], obbs[:, 0:3])
    ious = box3d_iou_batch(pred_bbox, ref_gt_bbox[scene_of])
    starts = np.concatenate([[0], np.cumsum(counts)])
    label_all = np.zeros(total, np.float32)
    keep, rows, cols = [], [], []
    srow = 0
    for i in range(batch_size):                       # B iterations of O(1) numpy on tiny slices
        n = counts[i]
        if n == 0:
            continue
        seg = ious[starts[i]:starts[i + 1]]
        label_all[starts[i] + int(seg.argmax())] = 1.0  # the box with the highest IoU is the positive
        if n >= 2:
            rows.append(np.full(n, srow))
            cols.append(np.arange(n))
            keep.append(1.0 if seg.max() >= 0.2 else 0.0)
            srow += 1
    out.update(starts=starts, srow=srow)
    device = torch.device(device)
    pin = device.type == "cuda"
    if srow:
        scored = np.concatenate([np.arange(starts[i], starts[i + 1]) for i in range(batch_size) if counts[i] >= 2])
        lmax = max(c for c in counts if c >= 2)
        flat_h = np.concatenate(rows) * lmax + np.concatenate(cols)
        lab_h = label_all[scored]
    else:
        lmax, flat_h, lab_h = 0, np.zeros(0, np.int64), np.zeros(0, np.float32)
    fh = torch.empty(total + lab_h.size + srow, dtype=torch.float32, pin_memory=pin)
    fv = fh.numpy()
    fv[:total] = label_all
    fv[total:total + lab_h.size] = lab_h
    fv[total + lab_h.size:] = np.asarray(keep, np.float32)
    al = out.pop("area_label_host", None)
    nal = 0 if al is None else al.size
    seg_h = np.concatenate([[0], np.cumsum([c for c in counts if c >= 2])]).astype(np.int64)   # scored-scene offsets
    ih = torch.empty(flat_h.size + nal + seg_h.size, dtype=torch.int64, pin_memory=pin)
    ih.numpy()[:flat_h.size] = flat_h
    if nal:
        ih.numpy()[flat_h.size:flat_h.size + nal] = al
    ih.numpy()[flat_h.size + nal:] = seg_h
    fbuf = fh.to(device, non_blocking=True)
    buf = ih.to(device, non_blocking=True)
    out.update(lmax=lmax, fbuf=fbuf, buf=buf, label_dev=fbuf[:total], lab=fbuf[total:total + lab_h.size],
               keep_dev=fbuf[total + lab_h.size:], flat=buf[:flat_h.size],
               area_label=buf[flat_h.size:flat_h.size + nal] if nal else None, seg_off=buf[flat_h.size + nal:])
    return out


import os as _os
FUSED_LOSS = _os.environ.get('IRX_FUSED_LOSS', '1') != '0'     # dev / test switch: 0 = the operator-by-operator formulation


def get_loss(data_dict, config):
    """Same outputs as the reference (loss, ref_loss, lang_loss, seg_loss, seg_acc, cluster_label), but batched:
    IoU labelling = one vectorised numpy pass over all candidates (prepare_labels), ONE H2D copy of the labels, and the
    per-sample ContrastiveLoss evaluated for all scenes at once on a (scenes x max_candidates) padded matrix (-inf
    padding for the log-sum-exp) — ~10 launches instead of ~10 per sample."""
    dev = data_dict["lang_scores"].device
    lp = data_dict.pop('_loss_prepared', None)
    if lp is None:
        lp = prepare_labels(data_dict, config, dev)
    if (FUSED_LOSS and dev.type == 'cuda' and lp.get('area_label') is not None and lp['total'] > 0 and lp['srow'] > 0
            and data_dict['lang_scores'].dim() == 2 and data_dict['seg_scores'].dim() == 2):
        # the whole loss in one launch each way (csrc/irx_match.hip, k_total_loss): it sits on the step's critical path between
        # the last head's forward and the first head's backward
        from . import heads
        from .dense import TotalLossFn
        largs = (data_dict['lang_scores'], data_dict['seg_scores'], data_dict['attribute_scores'], data_dict['relation_scores'],
                 data_dict['scene_scores'], data_dict['object_cat'].to(dev), lp['area_label'], lp['lab'], lp['seg_off'], lp['keep_dev'],
                 5.0, 0.2, 10.0, lp['batch_size'])
        out = heads.total_loss(*largs)               # the C++ node (csrc/heads_nodes.cpp); None: the Python node
        loss, ref_loss, lang_loss, seg_loss, seg_acc = out if out is not None else TotalLossFn.apply(*largs)
        starts, label_dev, counts = lp['starts'], lp['label_dev'], lp['counts']
        data_dict.update(lang_loss=lang_loss, ref_loss=ref_loss, loss=loss, seg_loss=seg_loss, seg_acc=seg_acc,
                         cluster_label=[label_dev[starts[i]:starts[i + 1]] if counts[i] else [] for i in range(lp['batch_size'])])
        data_dict['_labels'] = lp
        return data_dict
    lang_loss = compute_lang_classification_loss(data_dict)
    data_dict["lang_loss"] = lang_loss
    if lp.get('area_label') is not None:             # label computed with the other labels on the host
        pred, label = data_dict['seg_scores'], lp['area_label']
        seg_loss = nn.functional.cross_entropy(pred, label)
        seg_acc = (torch.argmax(pred, 1) == label).sum() / float(label.numel())
    else:
        seg_loss, seg_acc = compute_scene_mask_loss(data_dict)
    batch_size, counts, total, srow, lmax = lp['batch_size'], lp['counts'], lp['total'], lp['srow'], lp['lmax']
    margin, gamma = 0.2, 5.0
    if total == 0:
        ref_loss = torch.zeros(1, device=dev)
        cluster_label = [[] for _ in range(batch_size)]
    else:
        starts, label_dev = lp['starts'], lp['label_dev']
        cluster_label = [label_dev[starts[i]:starts[i + 1]] if counts[i] else [] for i in range(batch_size)]
        if srow == 0:
            ref_loss = torch.zeros(1, device=dev)
        else:
            flat, lab, keep_dev = lp['flat'], lp['lab'], lp['keep_dev']
            if dev.type == 'cuda':                     # all scenes in one launch each way (csrc/irx_match.hip)
                from .dense import ContrastiveFn
                ref_loss = ContrastiveFn.apply(data_dict['attribute_scores'], data_dict['relation_scores'],
                                               data_dict['scene_scores'], lab, lp['seg_off'], keep_dev, gamma, margin)
            else:                                      # host tensors (CPU tests): padded-matrix formulation
                score = (data_dict['attribute_scores'] + data_dict['relation_scores'] + data_dict['scene_scores']) * gamma
                sim = torch.zeros(srow * lmax, dtype=score.dtype, device=dev).index_put((flat,), score * lab).view(
                    srow, lmax).sum(1)
                neg = torch.full((srow * lmax,), float("-inf"), dtype=score.dtype, device=dev).index_put(
                    (flat,), score * (1.0 - lab)).view(srow, lmax)
                per_scene = torch.clamp(torch.logsumexp(neg, dim=1) - sim + margin, min=0)
                ref_loss = (per_scene * keep_dev).sum().reshape(1)

    ref_loss = ref_loss / batch_size
    data_dict['ref_loss'] = ref_loss
    data_dict['loss'] = 10 * ref_loss + lang_loss + seg_loss
    data_dict['seg_loss'] = seg_loss
    data_dict['seg_acc'] = seg_acc
    data_dict['cluster_label'] = cluster_label
    data_dict['_labels'] = lp                          # get_eval reuses the resident label tensors / offsets
    return data_dict
