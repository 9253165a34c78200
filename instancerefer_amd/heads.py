"""One autograd node per matching head (csrc/heads_nodes.cpp; round 6).

The operator-by-operator heads (attribute_module.py / scene_module.py: ~30 autograd nodes, a dozen of them Python
autograd.Functions) cost the training thread ~0.5 ms in the forward and the autograd engine ~1.2 ms in the backward — with
both encoders idle on the GPU meanwhile (profiles/r05_i_timeline_bf16.txt). Here the SAME C-ABI calls are issued in the SAME
order from C++:

  scene_head()  SceneModule.head : BEV rows -> BatchNorm2d/ReLU -> Conv2d -> BatchNorm2d/ReLU -> Dropout -> Conv2d -> language
                                   attention pooling -> area classifier            (reference models/scene_module.py:61-96)
  attr_scene()  AttributeModule.forward + SceneModule.forward's scores : global max pooling -> MLPs -> cosine scores
                                                          (models/attribute_module.py:105-126, models/scene_module.py:98-106)
  total_loss()  get_loss's arithmetic (lib/loss_helper.py:196-269)

Each returns None / False when its preconditions do not hold (eval mode, no C++ module, sync BatchNorm over several ranks, a
module layout other than the reference's, a one-sample batch) and the caller runs the per-operator path: same parameters, same
state-dict keys, bit-identical results with dropout off (tests/test_heads_gpu.py). IRX_FUSED_HEADS=0 switches all three off.
Dropout masks come from the library's counter-based hash (as in irx_mlp2_fwd), keyed by a seed drawn from torch's default
generator like every other dropout of this package."""
import os

import torch
import torch.nn as nn

from . import _lib
from .dense import _dropout_seed

FUSED = os.environ.get("IRX_FUSED_HEADS", "1") != "0"
AUX_WGRAD = os.environ.get("IRX_HEAD_AUX", "1") != "0"       # the scene head's weight gradients on a lent second stream (dev A/B; bit-identical)
PRE_LANG = os.environ.get("IRX_PRE_LANG", "1") != "0"        # the heads' language-side MLPs as nodes of their own behind the language module
CALLS = {"scene_head": 0, "attr_scene": 0, "total_loss": 0, "relation_head": 0, "lang_pool": 0}          # how often each node was taken (tests assert the path)


def _mod():
    if not FUSED:
        return None
    from . import _nodes
    m = _nodes.load()
    return m if (m is not None and hasattr(m, "scene_head")) else None


def _i64(seed):
    return seed if seed < (1 << 63) else seed - (1 << 64)


def _multi_rank_sync(bn):
    if not getattr(bn, "_irx_sync", False):
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _bn_ok(bn, cls):
    return (isinstance(bn, cls) and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
            and not _multi_rank_sync(bn))


def _mlp_ok(seq, norm, dropout):
    """nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear) exactly as the reference builds it"""
    mods = list(seq)
    if len(mods) != (5 if dropout else 4):
        return False
    if not (isinstance(mods[0], nn.Linear) and isinstance(mods[2], nn.ReLU) and isinstance(mods[-1], nn.Linear)
            and mods[0].bias is not None and mods[-1].bias is not None):
        return False
    if dropout and not isinstance(mods[3], nn.Dropout):
        return False
    if norm == "bn":
        return _bn_ok(mods[1], nn.BatchNorm1d)
    ln = mods[1]
    return isinstance(ln, nn.LayerNorm) and ln.elementwise_affine and len(ln.normalized_shape) == 1 and ln.bias is not None


def _mlp_params(seq):
    return [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias, seq[-1].weight, seq[-1].bias]


def _sink(key, params):
    """(slot vector, keep-alive tensors) of optim.FlatAdam's native gradient sink for `params`, or ((), ())"""
    if not torch.is_grad_enabled():
        return (), ()
    sink = getattr(params[0], "_irx_sink", None)
    if sink is None or any(getattr(p, "_irx_sink", None) is None or p._irx_sink[0] is not sink[0] for p in params):
        return (), ()
    ent = sink[0].native_sink(key, params)
    return ent if ent is not None else ((), ())


def _drop_p(m):
    return float(m.p) if m.training else 0.0


# ----------------------------------------------------------------------------------------------------- per-forward seeds
SEED_NAMES = ("rel_lang", "rel_vis", "scene_conv", "scene_lang", "fc1")


def draw_seeds(data_dict, device):
    """Every dropout seed of the fused heads for ONE forward, drawn on the calling (training) thread in a fixed order — the heads run
    on three streams and two threads, and seeds drawn where they are used would depend on which thread reaches the generator first
    (ADVICE r5). Stored as data_dict['_seeds']; a head that finds none draws its own."""
    data_dict['_seeds'] = {n: _dropout_seed(device) for n in SEED_NAMES}


def _seed(data_dict, name, device, p):
    if p <= 0:
        return 0
    seeds = data_dict.get('_seeds')
    return seeds[name] if seeds is not None else _dropout_seed(device)


# ------------------------------------------------------------------- the heads' language-side MLPs, ahead of the heads
def pre_lang_ok(model, data_dict):
    """The scene head's and the attribute head's lang_emb_fc depend on the language module only: run as nodes of their own right
    behind it (on the language stream, by the helper thread) they leave both heads' chains — 2 launches forward and 2-3 backward per
    head that sat between an encoder and its backward. Only when both fused heads will run (training, the reference's layout)."""
    if not PRE_LANG or _mod() is None or not (model.training and torch.is_grad_enabled()):
        return False
    a = model.args
    if not (a.attribute_module and a.scene_module and hasattr(model.scene, 'head')):
        return False
    lang = data_dict['lang_feat']
    prep = data_dict.get('_attr_prepared')
    return (lang.is_cuda and lang.shape[0] >= 2 and prep is not None and prep[0] is not None
            and _mlp_ok(model.scene.lang_emb_fc, "ln", True) and _mlp_ok(model.attribute.lang_emb_fc, "bn", False))


class PreLang:
    """callable(data_dict) -> data_dict with '_scene_lang_h' / '_attr_lang_h' (dense.mlp2: one C++ node each)"""

    def __init__(self, model):
        self.scene_fc, self.attr_fc = model.scene.lang_emb_fc, model.attribute.lang_emb_fc

    def __call__(self, data_dict):
        from .dense import mlp2
        dev = data_dict['lang_scene_feats'].device
        data_dict['_scene_lang_h'] = mlp2(self.scene_fc, data_dict['lang_scene_feats'],
                                          seed=_seed(data_dict, "scene_lang", dev, _drop_p(self.scene_fc[3])))
        data_dict['_attr_lang_h'] = mlp2(self.attr_fc, data_dict['lang_attr_feats'])
        return data_dict


# ---------------------------------------------------------------------------------------------------------------- scene head
def scene_head_ok(sm, feats, lang_feats):
    mod = _mod()
    if mod is None or not (sm.training and torch.is_grad_enabled() and feats.F.is_cuda and feats.F.dtype == torch.float32):
        return None
    from .basic_blocks import ToDenseBEVConvolution
    tb, ve = sm.to_bev, sm.vis_emb_fc
    ok = (len(tb) == 4 and isinstance(tb[1], ToDenseBEVConvolution) and _bn_ok(tb[2], nn.BatchNorm2d) and isinstance(tb[3], nn.ReLU)
          and len(ve) == 5 and isinstance(ve[0], nn.Conv2d) and _bn_ok(ve[1], nn.BatchNorm2d) and isinstance(ve[2], nn.ReLU)
          and isinstance(ve[3], nn.Dropout) and isinstance(ve[4], nn.Conv2d)
          and all(c.kernel_size[0] == c.kernel_size[1] and c.stride == (1, 1) and c.padding == (0, 0) and c.groups == 1
                  and c.dilation == (1, 1) and c.bias is not None for c in (ve[0], ve[4]))
          and _mlp_ok(sm.lang_emb_fc, "ln", True) and _mlp_ok(sm.cls, "bn", False)
          and lang_feats.shape[0] >= 2 and sm.h_dim <= 256)
    return mod if ok else None


def scene_head(sm, feats, data_dict):
    """SceneModule.head from the encoder's output SparseTensor on: fills vis_atten, seg_scores, _scene_feats. -> True, or False
    when the per-operator path has to run."""
    lang_feats = data_dict['lang_scene_feats']
    mod = scene_head_ok(sm, feats, lang_feats)
    if mod is None:
        return False
    pre = data_dict.pop('_scene_lang_h', None)       # lang_emb_fc already applied (PreLang), or None: the node runs it
    from .basic_blocks import _grid_tables
    batch_size = data_dict['point_min'].shape[0]
    x = feats.canonical()
    lv = x.level()
    bev = sm.to_bev[1]
    nx, ny = bev.bev_shape
    nz = bev.n_kernels
    tbl, cell, zbin = lv.bev(nx, ny, nz)
    tbl_t = lv.bev_t(nx, ny, nz)                     # (both usually built by the preparation stage already)
    ncell = lv.batch_size * nx * ny
    if lv.batch_size != batch_size or x.F.shape[0] == 0:
        return False
    ve = sm.vis_emb_fc
    k0, k1 = ve[0].kernel_size[0], ve[4].kernel_size[0]
    f0, b0, n_out0, n_in0 = _grid_tables(batch_size, nx, ny, k0, x.F.device)
    f1, b1, n_out1, n_in1 = _grid_tables(batch_size, nx - k0 + 1, ny - k0 + 1, k1, x.F.device)
    cache = sm.__dict__.get('_irx_head_params')
    if cache is None:
        bn0, bn1, cls = sm.to_bev[2], ve[1], sm.cls
        base = [bev.kernel, bn0.weight, bn0.bias, ve[0].weight, ve[0].bias, bn1.weight, bn1.bias, ve[4].weight, ve[4].bias]
        stats = [bn0.running_mean, bn0.running_var, bn1.running_mean, bn1.running_var, cls[1].running_mean, cls[1].running_var]
        counters = [b.num_batches_tracked for b in (bn0, bn1, cls[1]) if b.num_batches_tracked is not None]
        cache = sm.__dict__['_irx_head_params'] = (base + _mlp_params(sm.lang_emb_fc) + _mlp_params(cls), base + _mlp_params(cls),
                                                   stats, counters)
    params = cache[1] if pre is not None else cache[0]
    stats, counters = cache[2], cache[3]
    bn0, bn1 = sm.to_bev[2], ve[1]
    p_conv, p_lang = _drop_p(ve[3]), (0.0 if pre is not None else _drop_p(sm.lang_emb_fc[3]))
    dev = x.F.device
    s_conv = _seed(data_dict, "scene_conv", dev, p_conv)
    s_lang = _seed(data_dict, "scene_lang", dev, p_lang)
    f = [bn0.eps, bn0.momentum, bn1.eps, bn1.momentum, p_conv, sm.lang_emb_fc[1].eps, p_lang, sm.cls[1].eps, sm.cls[1].momentum]
    slots, keep = _sink(("scene_head", id(sm), pre is not None), params)
    aux = data_dict.get('_aux_stream')               # a second stream for the backward's weight gradients (InstanceRefer lends one)
    aux = [aux.cuda_stream, aux.stream_id, aux.device_index, aux.device_type] if (aux is not None and AUX_WGRAD) else []
    atten, seg, vec = mod.scene_head(x.F, pre if pre is not None else lang_feats, pre is not None, tbl, tbl_t, ncell, batch_size,
                                     [f0, b0, f1, b1], [n_out0, n_in0, n_out1, n_in1], params, stats, f, [_i64(s_conv), _i64(s_lang)],
                                     _lib.stream_ptr(), aux, list(slots), list(keep))
    if counters:
        from . import _counters
        _counters.bump(counters)
    h, w = nx - k0 - k1 + 2, ny - k0 - k1 + 2
    data_dict['vis_atten'] = atten.reshape(batch_size, h, w)
    data_dict['seg_scores'] = seg
    data_dict['_scene_feats'] = vec
    CALLS["scene_head"] += 1
    return True


# ------------------------------------------------------------------------------------------- attribute head + scene scores
def attr_head_ok(am, sm, data_dict):
    mod = _mod()
    if mod is None or not hasattr(mod, "cosine_rows") or not (am.training and sm.training and torch.is_grad_enabled()):
        return None
    prep = data_dict.get('_attr_prepared')
    if prep is None or prep[0] is None:
        return None
    lang = data_dict['lang_attr_feats']
    ok = (lang.is_cuda and lang.shape[0] >= 2 and _mlp_ok(am.lang_emb_fc, "bn", False) and _mlp_ok(am.vis_emb_fc, "ln", False)
          and _mlp_ok(sm.vis_emb_fc1, "ln", True))
    return mod if ok else None


def attr_head(am, sm, data_dict):
    """AttributeModule.forward plus the candidate side of SceneModule.forward's scores (vis_emb_fc1) as one node: fills
    num_filtered_objs, pred_obb_batch, obj_feats, attribute_scores, _sel_dev and '_obj_h' (what scene_scores() takes against the scene
    vector). Needs the prepared candidates (InstanceRefer.prepare) but NOT the scene head: it runs beside it, forward and backward.
    -> True, or False when the per-operator path has to run."""
    mod = attr_head_ok(am, sm, data_dict)
    if mod is None:
        return False
    from .data import selection_on_device, upload_instances
    from .sparse.encoder_fn import lane_of, lane_wait
    st, sel = data_dict.pop('_attr_prepared')
    data_dict['num_filtered_objs'] = sel['num_filtered_objs']
    data_dict['pred_obb_batch'] = sel['pred_obb_batch']
    feats = data_dict.pop('_attr_encoded', None)
    if feats is None:
        feats = am.net(st)
    elif hasattr(feats, 'attach'):
        feats = feats.attach()
    lane_wait(lane_of(am.net))                      # the encoder may be issued by a library thread
    x = feats.canonical()
    lv = x.level()
    dev = x.F.device
    sd = selection_on_device(sel, upload_instances(data_dict), dev)
    data_dict['_sel_dev'] = sd
    pre = data_dict.pop('_attr_lang_h', None)
    cache = am.__dict__.get('_irx_head_params')
    if cache is None or cache[4] is not sm:
        tail = _mlp_params(am.vis_emb_fc) + _mlp_params(sm.vis_emb_fc1)
        bn = am.lang_emb_fc[1]
        cache = am.__dict__['_irx_head_params'] = (_mlp_params(am.lang_emb_fc) + tail, tail, [bn.running_mean, bn.running_var],
                                                   [bn.num_batches_tracked] if bn.num_batches_tracked is not None else [], sm)
    params = cache[1] if pre is not None else cache[0]
    stats, counters = cache[2], ([] if pre is not None else cache[3])
    bn = am.lang_emb_fc[1]
    p_fc1 = _drop_p(sm.vis_emb_fc1[3])
    s_fc1 = _seed(data_dict, "fc1", dev, p_fc1)
    f = [bn.eps, bn.momentum, am.vis_emb_fc[1].eps, sm.vis_emb_fc1[1].eps, p_fc1, 1e-12]
    slots, keep = _sink(("attr_head", id(am), pre is not None), params)
    obj, s_attr, obj_h = mod.attr_head(x.F, lv.offsets(), lv.batch_size, sd['cand_scene'], pre if pre is not None else data_dict['lang_attr_feats'],
                                       pre is not None, params, stats, f, [_i64(s_fc1)], _lib.stream_ptr(), list(slots), list(keep))
    if counters:
        from . import _counters
        _counters.bump(counters)
    data_dict['obj_feats'] = obj
    data_dict['attribute_scores'] = s_attr
    data_dict['_obj_h'] = obj_h
    CALLS["attr_scene"] += 1
    return True


def scene_scores(data_dict):
    """cosine of every candidate's vis_emb_fc1 vector ('_obj_h', attr_head) against its scene's vector ('_scene_feats', scene head):
    SceneModule.forward's score (reference models/scene_module.py:104-106) as a C++ node."""
    mod = _mod()
    obj_h = data_dict.pop('_obj_h')
    vec = data_dict.pop('_scene_feats')
    data_dict['scene_scores'] = mod.cosine_rows(obj_h, vec, data_dict['_sel_dev']['cand_scene'], 1e-8, _lib.stream_ptr())
    return data_dict


# ---------------------------------------------------------------------------------------------------------------------- loss
def total_loss(lang_scores, seg_scores, s1, s2, s3, lang_label, seg_label, lab, seg_off, keep, gamma, margin, ref_weight, batch_size):
    """-> (loss (1,), ref_loss (1,), lang_loss (), seg_loss (), seg_acc ()) through the C++ node, or None"""
    mod = _mod()
    if mod is None or not lang_scores.is_cuda:
        return None
    CALLS["total_loss"] += 1
    return mod.total_loss(lang_scores, seg_scores, s1, s2, s3, lang_label, seg_label, lab, seg_off, keep, float(gamma), float(margin),
                          float(ref_weight), int(batch_size), _lib.stream_ptr())


# ------------------------------------------------------------------------------------------------------------ relation head
def relation_head(rm, lang_feats, prep, data_dict):
    """RelationModule.forward behind the prepared node features: language MLP -> edge convolution over the prepared kNN grid -> visual
    MLP -> cosine score, as one node (reference models/relation_module.py:84-107). Fills relation_scores. -> True / False."""
    mod = _mod()
    if (mod is None or not hasattr(mod, "relation_head") or not (rm.training and torch.is_grad_enabled() and lang_feats.is_cuda)
            or prep is None or len(prep) < 5 or lang_feats.shape[0] < 2):
        return False
    sel, sd, centres, feats, nbr = prep[:5]
    gcn = rm.gcn
    if not (_mlp_ok(rm.lang_emb_fc, "bn", True) and _mlp_ok(rm.vis_emb_fc, "ln", True) and gcn.fused_supported(feats.shape[1])
            and feats.dtype == torch.float32 and nbr.shape[0] > 0):
        return False
    cache = rm.__dict__.get('_irx_head_params')
    if cache is None:
        params = _mlp_params(rm.lang_emb_fc) + _mlp_params(rm.vis_emb_fc)
        params += [gcn.weight[0].weight, gcn.weight[0].bias, gcn.weight[2].weight, gcn.weight[2].bias,
                   gcn.mlp[0].weight, gcn.mlp[0].bias, gcn.mlp[2].weight, gcn.mlp[2].bias]
        bn = rm.lang_emb_fc[1]
        cache = rm.__dict__['_irx_head_params'] = (params, [bn.running_mean, bn.running_var],
                                                   [bn.num_batches_tracked] if bn.num_batches_tracked is not None else [])
    params, stats, counters = cache
    bn = rm.lang_emb_fc[1]
    dev = lang_feats.device
    p_lang, p_vis = _drop_p(rm.lang_emb_fc[3]), _drop_p(rm.vis_emb_fc[3])
    s_lang = _seed(data_dict, "rel_lang", dev, p_lang)
    s_vis = _seed(data_dict, "rel_vis", dev, p_vis)
    f = [bn.eps, bn.momentum, p_lang, rm.vis_emb_fc[1].eps, p_vis, 1e-8]
    slots, keep = _sink(("relation_head", id(rm)), params)
    (scores,) = mod.relation_head(lang_feats, feats, centres, sd['query_in_support'], nbr, sd['cand_scene'], gcn.num_classes, params, stats,
                                  f, [_i64(s_lang), _i64(s_vis)], _lib.stream_ptr(), list(slots), list(keep))
    if counters:
        from . import _counters
        _counters.bump(counters)
    data_dict['relation_scores'] = scores
    CALLS["relation_head"] += 1
    return True


# ----------------------------------------------------------------------------------------- language module: attention pooling
def lang_pool(lm, feats, embed, length):
    """The four attention heads of the language module as one node whose pooled vectors come out as four contiguous tensors (no
    select nodes behind them; reference models/lang_module.py:61-83). -> (att (B, T, 4), [attr, cls, rel, scene] each (B, E)) or None"""
    mod = _mod()
    if mod is None or not hasattr(mod, "lang_pool") or not (torch.is_grad_enabled() and feats.is_cuda):
        return None
    params = [lm.fc_a.weight, lm.fc_a.bias, lm.fc_cls.weight, lm.fc_cls.bias, lm.fc_rel.weight, lm.fc_rel.bias,
              lm.fc_scene.weight, lm.fc_scene.bias]
    slots, keep = _sink(("lang_pool", id(lm)), params)
    out = mod.lang_pool(feats, embed, length, params, _lib.stream_ptr(), list(slots), list(keep))
    CALLS["lang_pool"] += 1
    return out[0], out[1:]
