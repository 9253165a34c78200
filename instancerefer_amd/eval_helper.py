"""get_eval — drop-in for the reference's lib/eval_helper.py:11-113: predicted box = the candidate with the highest
summed score (the only candidate / a zero box when a scene has 1 / 0 candidates), IoU against the GT box, `ref_acc`
(arg-max hit for >= 2 candidates, IoU > 0.25 otherwise), Acc@0.25 / Acc@0.5 rates and the unique/multiple, "others"
masks. Same keys and list/array types as the reference; the per-sample device round trips (`.item()`, tensor slicing and
arg-max per scene) are replaced by ONE launch (irx_eval_select: per-scene arg-max of the summed scores and of the labels,
the chosen box and its float64 IoU, bit-identical to the numpy evaluation) and ONE D2H copy of B x 10 doubles when get_loss
left its resident label tensors behind; on host tensors (CPU tests) one padded arg-max + numpy."""
import numpy as np
import torch

from .loss_helper import _host_np, box3d_iou_batch, get_3d_box_batch


def construct_bbox_corners(center, box_size):
    """(3,), (3,) -> (8, 3) corners in the reference's order (utils/util.py:20-31)."""
    sx, sy, sz = box_size
    x = np.array([sx, sx, -sx, -sx, sx, sx, -sx, -sx]) / 2 + center[0]
    y = np.array([sy, -sy, -sy, sy, sy, -sy, -sy, sy]) / 2 + center[1]
    z = np.array([sz, sz, sz, sz, -sz, -sz, -sz, -sz]) / 2 + center[2]
    return np.stack([x, y, z], 1)


def get_eval(data_dict, config):
    lang_scores = data_dict["lang_scores"]
    batch_size = lang_scores.shape[0]
    object_cat = data_dict["object_cat"].to(lang_scores.device)
    data_dict["lang_acc"] = (torch.argmax(lang_scores, dim=1) == object_cat).float().mean()

    pred_obb_batch = data_dict['pred_obb_batch']
    counts = [int(p.shape[0]) for p in pred_obb_batch]
    scored = [i for i in range(batch_size) if counts[i] >= 2]
    lp = data_dict.get('_labels')
    on_device = None
    if lp is not None and lp.get('dev') is not None and lp['counts'] == counts and data_dict['attribute_scores'].is_cuda:
        from . import _lib
        d = lp['dev']
        res = torch.empty((batch_size, 10), dtype=torch.float64, device=d['gt'].device)
        sc = [data_dict[k].detach().float().contiguous() for k in ('attribute_scores', 'relation_scores', 'scene_scores')]
        _lib.call("irx_eval_select", _lib.ptr(sc[0]), _lib.ptr(sc[1]), _lib.ptr(sc[2]), _lib.ptr(lp['label_dev']),
                  _lib.ptr(d['obbs']), _lib.ptr(d['filtered']), _lib.ptr(d['starts']), _lib.ptr(d['scored_pos']),
                  _lib.ptr(d['gt']), batch_size, _lib.ptr(res), _lib.stream_ptr())
        on_device = res.cpu().numpy()                  # the ONE D2H copy of the evaluation
        scored = []
    # arg-max of the summed scores and of the cluster label for every scored scene: one padded matrix, one D2H
    pred_idx, tgt_idx = {}, {}
    if scored:
        dev = data_dict['attribute_scores'].device
        score = (data_dict['attribute_scores'] + data_dict['relation_scores'] + data_dict['scene_scores']).detach()
        lmax = max(counts[i] for i in scored)
        rows = np.concatenate([np.full(counts[i], r) for r, i in enumerate(scored)])
        cols = np.concatenate([np.arange(counts[i]) for i in scored])
        from .data import idx_tensor
        flat = idx_tensor(rows * lmax + cols, dev)
        pad = torch.full((len(scored) * lmax,), float("-inf"), device=dev, dtype=score.dtype).index_put((flat,), score)
        labels = torch.cat([torch.as_tensor(data_dict['cluster_label'][i], device=dev).float() for i in scored])
        lpad = torch.full((len(scored) * lmax,), float("-inf"), device=dev).index_put((flat,), labels)
        both = torch.stack([pad.view(len(scored), lmax).argmax(1), lpad.view(len(scored), lmax).argmax(1)]).cpu().numpy()
        for r, i in enumerate(scored):
            pred_idx[i], tgt_idx[i] = int(both[0, r]), int(both[1, r])

    ref_gt_obb = config.param2obb_batch(_host_np(data_dict, "ref_center_label"),
                                        _host_np(data_dict, "ref_heading_class_label"),
                                        _host_np(data_dict, "ref_heading_residual_label"),
                                        _host_np(data_dict, "ref_size_class_label"),
                                        _host_np(data_dict, "ref_size_residual_label"))
    if on_device is not None:
        for i in range(batch_size):
            if counts[i] >= 2:
                pred_idx[i], tgt_idx[i] = int(on_device[i, 0]), int(on_device[i, 1])
        ious, chosen = on_device[:, 2].copy(), on_device[:, 3:10].copy()
    else:
        chosen = np.zeros((batch_size, 7))
        for i in range(batch_size):
            if counts[i] == 1:
                chosen[i] = pred_obb_batch[i][0]
            elif counts[i] >= 2:
                chosen[i] = pred_obb_batch[i][pred_idx[i]]
        ious = box3d_iou_batch(get_3d_box_batch(chosen[:, 3:6], chosen[:, 6], chosen[:, 0:3]),
                               get_3d_box_batch(ref_gt_obb[:, 3:6], ref_gt_obb[:, 6], ref_gt_obb[:, 0:3]))
    um = _host_np(data_dict, "unique_multiple") if "unique_multiple" in data_dict else np.zeros(batch_size, np.int64)
    cat = _host_np(data_dict, "object_cat")
    ref_acc, pred_bboxes, gt_bboxes, multiple, others = [], [], [], [], []
    for i in range(batch_size):
        if counts[i] >= 2:
            ref_acc.append(1. if tgt_idx[i] == pred_idx[i] else 0.)
        else:
            ref_acc.append(1. if ious[i] > 0.25 else 0.)
        pred_bboxes.append(construct_bbox_corners(chosen[i, 0:3], chosen[i, 3:6]))
        gt_bboxes.append(construct_bbox_corners(ref_gt_obb[i, 0:3], ref_gt_obb[i, 3:6]))
        multiple.append(int(um[i]))
        others.append(1 if int(cat[i]) == 17 else 0)
    ious = [float(v) for v in ious]
    data_dict['ref_acc'] = ref_acc
    data_dict["ref_iou"] = ious
    arr = np.array(ious)
    data_dict["ref_iou_rate_0.25"] = arr[arr >= 0.25].shape[0] / arr.shape[0]
    data_dict["ref_iou_rate_0.5"] = arr[arr >= 0.5].shape[0] / arr.shape[0]
    data_dict["ref_multiple_mask"] = multiple
    data_dict["ref_others_mask"] = others
    data_dict["pred_bboxes"] = pred_bboxes
    data_dict["gt_bboxes"] = gt_bboxes
    return data_dict
