"""Whole-encoder executor: the 13 Conv3d->BatchNorm(->+res)->ReLU groups of SparseConvEncoder / BEVEncoder
(reference models/basic_blocks.py:59-95,136-171) issued as ONE autograd node.

The per-layer path (sparse/nn.py: conv_bn_act) costs ~3 autograd nodes, ~10 tensor allocations and ~100 us of Python
per layer and direction; with two encoders that was ~10 ms of host time per step, i.e. the step was host-bound.
Here forward and backward are ONE C-ABI call each (irx_encoder_forward / irx_encoder_backward, include/irx.h): the
library walks a descriptor table and launches the same kernels in the same order as the per-layer path (bit-identical
results); Python only fills the table and allocates three arenas.

Layer list (index: conv, residual source): 0 stem | per stage s: 3s+1 down (2^3/2), 3s+2 res-a, 3s+3 res-b (+ out of 3s+1).
"""
import os
import weakref

import numpy as np
import torch

from .. import _lib
from . import functional as F_

_f32 = torch.float32
_PAIR = (32, 64, 128)


def _uses_pairs(L):
    """Weight gradient through the rulebook pair lists (k_wgrad_pairs): the MFMA channel counts, and the 128 leading
    channels of the multiview stem (C0 = 129..136 -> 32, csrc/irx_spconv.hip "wide stem")."""
    if L.cin in _PAIR and L.cout in _PAIR:
        return True
    return (not L.down) and L.K == 27 and L.cout == 32 and 128 < L.cin <= 136


class _Layer:
    __slots__ = ("conv", "bn", "lv_in", "lv_out", "K", "cin", "cout", "tbl", "ld", "n_in", "n_out", "down", "res")


def _skeleton(encoder):
    """(conv, bn, down, res) of the 13 layers — static per encoder instance."""
    sk = encoder.__dict__.get("_irx_skeleton")
    if sk is None:
        sk = [(encoder.stem[0].net[0], encoder.stem[0].net[1], False, -1)]
        for stage in (encoder.stage1, encoder.stage2, encoder.stage3, encoder.stage4):
            sk.append((stage[0].net[0], stage[0].net[1], True, -1))
            d = len(sk) - 1
            rb = stage[1]
            if len(rb.downsample) != 0:
                raise NotImplementedError("encoder executor: projection shortcuts are not part of the InstanceRefer encoders")
            sk.append((rb.net[0], rb.net[1], False, -1))
            sk.append((rb.net[3], rb.net[4], False, d))
        encoder.__dict__["_irx_skeleton"] = sk
    return sk


class Plan(list):
    """The layers of one encoder pass + what only depends on the coordinate levels (`pre`: descriptor columns, arena layout,
    workspace size per storage mode). Cached on the pass's finest level, so the preparation stage can build it ahead of the
    forward (prebuild) — the forward half of the bf16 step is host-paced, DESIGN.md section 5.
    The plan lives in that level's __dict__, so it must not own the level: its layers see the finest level through a weak
    proxy and `root` is a weak reference (level -> plan -> level would be a reference cycle per batch, i.e. ~200 objects and
    ~120 device tensors per step that only the cyclic collector frees — measured round 6: a 6-10 ms generation-2 pause every
    ~30 steps of the benchmark loop). Whoever runs the plan holds the level: launch() pins it on the Launched pass / the node."""
    __slots__ = ("pre", "root")


def build_plan(encoder, level0):
    """Bind the coordinate levels / tables of this batch to the layer skeleton (the pyramid is already built)."""
    cache = level0.__dict__.setdefault("_irx_plans", {})
    token = encoder.__dict__.setdefault("_irx_token", _Token())     # (id() of a collected encoder can be handed out again)
    plan = cache.get(token)
    if plan is not None:
        return plan
    layers = Plan()
    layers.pre = {}
    layers.root = weakref.ref(level0)
    level0.build_kmaps()                # every level's kernel map in one native call (no-op for the ones already built)
    lv = weakref.proxy(level0)
    for conv, bn, down, res in _skeleton(encoder):
        L = _Layer()
        L.conv, L.bn, L.lv_in, L.down, L.res = conv, bn, lv, down, res
        L.K, L.cin, L.cout = conv.kernel.shape
        if down:
            dm = lv.down()
            L.lv_out, L.tbl, L.ld = dm.out_level, dm.child, dm.ld
        else:
            L.lv_out = lv
            L.tbl, L.ld = lv.nbr27()
        L.n_in, L.n_out = lv.n, L.lv_out.n
        layers.append(L)
        lv = L.lv_out
    cache[token] = layers               # only a COMPLETE plan is ever cached (an exception above leaves nothing behind)
    return layers


class _Token:
    """Identity of an encoder instance in the per-level plan cache."""
    __slots__ = ("__weakref__",)


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


# field order of the descriptor table: include/irx.h, enum IRX_ENC_* (tests/test_abi_cpu.py checks the two agree)
ENC_FIELDS = ("K", "CIN", "COUT", "N_IN", "N_OUT", "RES", "TBL", "LD", "TBL_B", "LD_B", "FLIP_B", "PAIR_IN", "PAIR_OUT",
              "PAIR_COUNTS", "LD_PAIRS", "W", "GAMMA", "BETA", "RUNNING_MEAN", "RUNNING_VAR", "X", "C", "Y", "MEAN",
              "INVSTD", "DW", "DGAMMA", "DBETA", "GY", "STORE", "MODE", "ORDER", "PROF", "DC2", "WSTREAM", "WROWS")
_E = {n: i for i, n in enumerate(ENC_FIELDS)}
_NF = len(ENC_FIELDS)
_ALIGN = 64                                           # float32 elements (256 B)


def _up(n):
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def _up256(nbytes):
    return (nbytes + 255) // 256 * 256


def storage_bf16(need_input_grad=False):
    """bf16 STORAGE of the activations / gradients inside the executor (set_compute_dtype("bf16"), library mode 2)?
    Not when the caller wants the gradient of the input features (that path is fp32 only)."""
    return _lib.load().irx_get_compute_dtype() == 2 and not need_input_grad


def _static_template(encoder, layers, params):
    """Descriptor rows with everything that does not change from batch to batch (layer shapes, residual links, parameter
    and running-stat pointers, relative stats offsets), cached per encoder; rebuilt when the parameters were re-homed
    (FlatAdam moves them into its flat buffer once)."""
    # every cached pointer is part of the anchor: parameters AND running statistics (model.to(), .float(),
    # load_state_dict(assign=True) may move any of them on its own)
    anchor = tuple(p.data_ptr() for p in params) + tuple(b.data_ptr() for L in layers for b in (L.bn.running_mean, L.bn.running_var))
    cached = encoder.__dict__.get("_irx_template")
    if cached is not None and cached[0] == anchor:
        return cached[1:]
    nl = len(layers)
    t = np.zeros((nl, _NF), dtype=np.int64)
    f = np.zeros((nl, 2), dtype=np.float64)
    counters = []
    for i, L in enumerate(layers):
        bn = L.bn
        t[i, _E["K"]], t[i, _E["CIN"]], t[i, _E["COUT"]], t[i, _E["RES"]] = L.K, L.cin, L.cout, L.res
        t[i, _E["W"]] = params[3 * i].data_ptr()
        t[i, _E["GAMMA"]], t[i, _E["BETA"]] = params[3 * i + 1].data_ptr(), params[3 * i + 2].data_ptr()
        t[i, _E["RUNNING_MEAN"]], t[i, _E["RUNNING_VAR"]] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        t[i, _E["MEAN"]], t[i, _E["INVSTD"]] = 4 * (i * 256), 4 * (i * 256 + 128)       # relative to the stats tensor
        f[i] = (bn.eps, bn.momentum)            # momentum None (cumulative average) is refused by can_fuse
        if bn.num_batches_tracked is not None:
            counters.append(bn.num_batches_tracked)
    cout = t[:, _E["COUT"]].copy()
    wsize = t[:, _E["K"]] * t[:, _E["CIN"]] * cout
    # parameter-gradient arena layout (kernel, gamma, beta per layer), 256 B aligned
    sizes = np.stack([_up_np(wsize), _up_np(cout), _up_np(cout)], 1).reshape(-1)
    poffs = (np.concatenate([[0], np.cumsum(sizes)[:-1]]) * 4).reshape(nl, 3)
    ptotal = int(sizes.sum())
    encoder.__dict__["_irx_template"] = (anchor, t, f, counters, cout, poffs, ptotal)
    return t, f, counters, cout, poffs, ptotal


def _up_np(a):
    return (a + (_ALIGN - 1)) // _ALIGN * _ALIGN


# ---- asynchronous issue (irx_encoder_submit / irx_encoder_wait, include/irx.h) ---------------------------------------
# An encoder with `_irx_lane` set (InstanceRefer does that for its two encoders) has its forward — and, when the
# parameter gradients go to the optimizer's sink and no input gradient is wanted, its backward — issued by a library
# thread, so ~0.2 ms of launch issue per pass leaves the Python thread. Whoever enqueues dependent work on that stream
# (or records an event on it) calls lane_wait() first. IRX_ENCODER_ASYNC=0 switches it off.
ASYNC = os.environ.get("IRX_ENCODER_ASYNC", "1") != "0"
_HELD = {}            # lane -> device buffers that must outlive the queued jobs


# ---- sync BatchNorm inside the one-call executor (irx_encoder_forward_sync / _backward_sync, include/irx.h) ----------------
# The library calls back between a layer's statistics and apply pass; the fold over the ranks is torch.distributed's. The pass
# runs inline on the calling thread (no lane): the callback needs the interpreter and the collective's stream is torch's current one.
SYNC_IN_EXECUTOR = os.environ.get("IRX_SYNC_BN_EXECUTOR", "1") != "0"
SYNC_STRIDE = 264                                      # IRX_ENC_SYNC_STRIDE
SYNC_CALLS = [0, 0]                                    # forward / backward passes that went through the sync executor (tests)
_SYNC_CTX = {}
_SYNC_ERR = []


def _allreduce_cb(user, buf, n, is_double, stream):
    try:
        import torch.distributed as dist
        sums, gsums, group = _SYNC_CTX[int(user)]
        t = sums if is_double else gsums
        off = (int(buf) - t.data_ptr()) // t.element_size()
        dist.all_reduce(t.view(-1)[off:off + int(n)], op=dist.ReduceOp.SUM, group=group)
        return 0
    except Exception as e:                             # never let an exception cross the C frame
        _SYNC_ERR.append(repr(e))
        return 1


import ctypes as _ct
_ALLREDUCE_C = _ct.CFUNCTYPE(_ct.c_int, _ct.c_void_p, _ct.c_void_p, _ct.c_int, _ct.c_int, _ct.c_void_p)(_allreduce_cb)


def _sync_group_for(layers):
    """The process group the encoder's BatchNorm layers fold their statistics over, or None (not converted / one rank)."""
    if not SYNC_IN_EXECUTOR or not any(getattr(L.bn, "_irx_sync", False) for L in layers):
        return None
    return F_.sync_group()


def lane_of(encoder):
    return encoder.__dict__.get("_irx_lane") if ASYNC else None


def lane_wait(lane):
    """Block until the lane has enqueued everything submitted to it; raise if one of its passes failed."""
    if lane is None:
        return
    rc = _lib.load().irx_encoder_wait(lane)
    _HELD.pop(lane, None)
    if rc:
        _lib.check(rc, "irx_encoder (asynchronous pass)")


# Launch order of k_spconv2's output tiles (csrc/irx_sched.hip): on for the stride-1 layers of levels with >= 512 tiles (more
# than one round of workgroups on 256 CUs x 2); IRX_TILE_ORDER=0 switches it off (dev A/B; results are bit-identical).
TILE_ORDER = os.environ.get("IRX_TILE_ORDER", "1") != "0"
TILE_ORDER_MIN_ROWS = int(os.environ.get("IRX_TILE_ORDER_MIN_ROWS", str(512 * 64)))

# The backward pass's weight gradients on a second stream beside the BatchNorm-backward / data-gradient chain (IRX_ENC_DC2 /
# IRX_ENC_WSTREAM, include/irx.h; bit-identical) — for an encoder whose owner lends a stream (`_irx_wgrad_stream`: InstanceRefer's
# three-stream forward lends the language stream to the scene encoder). IRX_WGRAD_STREAM=0 keeps the one-stream order.
WGRAD_STREAM = os.environ.get("IRX_WGRAD_STREAM", "1") != "0"
WGRAD_STREAM_ROWS = int(os.environ.get("IRX_WGRAD_STREAM_ROWS", "30000"))    # only layers with fewer output rows (0: all): IRX_ENC_WROWS

# Tests only (tests/test_bf16_gpu.py): a dict that receives the executor's arenas, so that every layer's stored tensors
# (conv output c_i, layer output y_i, gradient in flight gy_i) can be compared one layer at a time; None = no tracing.
TRACE = None


def trace_tensors(tr):
    """TRACE dict -> {"x": [...], "c": [...], "y": [...], "mean", "invstd", "gy": [...]} as float32 tensors on the device
    (bf16-stored tensors widened exactly), one entry per layer; call after a device sync."""
    f = tr["fwd"]
    nl, n_out, cout = len(f["layers"]), f["n_out"], f["cout"]

    def view(buf, off, n, c, bf):
        if bf:
            return buf[off:off + 2 * n * c].view(torch.bfloat16).view(n, c).float()
        return buf[off:off + 4 * n * c].view(_f32).view(n, c).clone()
    out = {"c": [], "y": [], "x": [f["x0"]], "mean": [], "invstd": [], "gy": []}
    for i in range(nl):
        n, c = int(n_out[i]), int(cout[i])
        out["c"].append(view(f["arena"], int(f["start"][i]), n, c, f["store"]))
        out["y"].append(view(f["arena"], int(f["start"][i] + f["cb"][i]), n, c, f["store"] and i < nl - 1))
        out["mean"].append(f["stats"][i, 0, :c].clone())
        out["invstd"].append(f["stats"][i, 1, :c].clone())
    out["x"] += out["y"][:-1]
    b = tr.get("bwd")
    if b is not None:
        for i in range(nl - 1):
            out["gy"].append(view(b["garena"], int(b["goffs"][i]), int(n_out[i]), int(cout[i]), b["store"]))
        out["gy"].append(b["dout"])
    return out


def _level_desc(encoder, layers, params, store):
    """Everything of the forward descriptor that depends on the coordinate levels, the parameters' addresses and the compute
    mode only — not on this pass's activations: the table with sizes / table pointers / launch orders filled in and the arena
    offsets in the C / Y columns, the arena layout and the workspace size. Cached on the plan per (storage, mode)."""
    lib = _lib.load()
    mode = int(lib.irx_get_compute_dtype())
    tmpl, fdesc, counters, cout, poffs, ptotal = _static_template(encoder, layers, params)
    # (the spconv3 / wgrad3 knobs change the workspace size and the split counts: part of the key)
    key = (bool(store), mode, id(tmpl), int(lib.irx_debug_get_knob(b"spconv3")), int(lib.irx_debug_get_knob(b"wgrad3")))
    hit = layers.pre.get(key) if isinstance(layers, Plan) else None
    if hit is not None:
        return hit
    nl = len(layers)
    n_out = np.fromiter((L.n_out for L in layers), dtype=np.int64, count=nl)
    desc = tmpl.copy()
    desc[:, _E["N_IN"]] = np.fromiter((L.n_in for L in layers), dtype=np.int64, count=nl)
    desc[:, _E["N_OUT"]] = n_out
    desc[:, _E["TBL"]] = np.fromiter((L.tbl.data_ptr() for L in layers), dtype=np.int64, count=nl)
    desc[:, _E["LD"]] = np.fromiter((L.ld for L in layers), dtype=np.int64, count=nl)
    # activation arena (bytes): conv output c_i and layer output y_i of every layer; with bf16 storage every tensor
    # but the last layer's output is 2 bytes per element
    esz = np.full(nl, 2 if store else 4, dtype=np.int64)
    ysz = esz.copy()
    ysz[-1] = 4
    cb, yb = _up256(n_out * cout * esz), _up256(n_out * cout * ysz)
    start = np.concatenate([[0], np.cumsum(cb + yb)[:-1]])
    total = int((cb + yb).sum())
    desc[:, _E["STORE"]] = int(store)
    desc[:, _E["MODE"]] = mode                   # pinned for this pass and its backward (include/irx.h)
    if TILE_ORDER:                               # heaviest output tiles first on the levels with more than one round of tiles
        desc[:, _E["ORDER"]] = np.fromiter((L.lv_in.order27().data_ptr() if (not L.down and L.n_out >= TILE_ORDER_MIN_ROWS and
                                                                             L.cin in _PAIR and L.cout in _PAIR) else 0
                                            for L in layers), dtype=np.int64, count=nl)      # (k_spconv2 layers only)
    desc[:, _E["C"]] = start
    desc[:, _E["Y"]] = start + cb
    nbytes = lib.irx_encoder_workspace_bytes(desc.ctypes.data, fdesc.ctypes.data, nl, 0)
    out = (desc, fdesc, counters, cout, poffs, ptotal, n_out, cb, start, total, nbytes)
    if isinstance(layers, Plan):
        layers.pre[key] = out
    return out


def prebuild(encoder, level0):
    """Preparation stage (any thread): the plan of `encoder` on this batch's levels and its level-only descriptor, so that the
    forward only allocates, adds addresses and submits."""
    if not can_fuse(encoder):
        return
    layers = build_plan(encoder, level0)
    params = []
    for L in layers:
        params += [L.conv.kernel, L.bn.weight, L.bn.bias]
    _level_desc(encoder, layers, params, storage_bf16(False))


def _backward_tables(layers):
    """The tables only the backward pass needs: pair lists of every layer whose weight gradient runs over them (all missing ones
    in ONE library call) and the transposed child maps of the strided layers. Lazily built and cached on the levels."""
    missing, seen = [], set()
    for L in layers:
        if _uses_pairs(L):
            if L.down:
                dm = L.lv_in.down()
                if dm._pairs is None and id(dm) not in seen:
                    seen.add(id(dm))
                    missing.append((dm, (dm.child, dm.ld, dm.out_level.n, 8)))
            elif L.lv_in._pairs27 is None and id(L.lv_in) not in seen:
                seen.add(id(L.lv_in))
                tbl27, ld27 = L.lv_in.nbr27()
                missing.append((L.lv_in, (tbl27, ld27, L.lv_in.n, 27)))
    if missing:
        for (holder, _), built in zip(missing, F_.pairs_build_multi([m[1] for m in missing])):
            if hasattr(holder, "_pairs27"):
                holder._pairs27 = built
            else:
                holder._pairs = built
    for L in layers:
        if L.down:
            L.lv_in.down().child_t()


def prebuild_backward(encoder, level0, use_stream=None):
    """Build the backward-only tables of `encoder`'s pass over this pyramid NOW, on the current stream (~70 us of kernels for the
    scene encoder at B = 16 that otherwise sit at the head of its backward, on the step's critical path): InstanceRefer's
    multi-stream forward calls it behind the scene head, when that stream has nothing else to do until the loss comes back.
    use_stream: the stream the backward will run on when it is not the current one (an event orders it behind the build)."""
    if not can_fuse(encoder):
        return
    layers = build_plan(encoder, level0)
    _backward_tables(layers)
    if use_stream is not None and use_stream != torch.cuda.current_stream():
        ev = torch.cuda.Event()
        ev.record()
        layers.pre["bwd_event"] = ev
        for L in layers:
            if _uses_pairs(L):
                for t in (L.lv_in.down().pairs() if L.down else L.lv_in.pairs27())[:3]:
                    t.record_stream(use_stream)
            if L.down:
                L.lv_in.down().child_t()[0].record_stream(use_stream)


class Launched:
    """An encoder forward pass that has been ISSUED (kernels enqueued or handed to a lane) but whose autograd node does not exist
    yet: launch() -> Launched, EncoderFn.apply(feats, encoder, launched, *params) creates the node later without launching
    anything. Why: the autograd engine runs ready nodes in the reverse order of their CREATION. The scene encoder must be issued
    first in the forward (it is the long pole) — as an ordinary node it would then be the LAST one the backward reaches, after
    every head and the candidate encoder (measured, round 5: its backward started 1.0 ms after the candidate encoder's, the side
    stream idle meanwhile). Created at the head of SceneModule.forward instead, it is replayed right behind the scene head."""
    __slots__ = ("layers", "desc", "fdesc", "extra", "store", "prof", "lane", "sync", "sink", "saved", "out", "need_dx0", "level0")


def launch(feats, encoder, layers, params, need_dx0=False):
    """Issue the forward pass (one irx_encoder_forward call / lane submission over a descriptor table) -> Launched."""
    lib = _lib.load()
    dev = feats.device
    x0 = feats.detach().contiguous().float()
    nl = len(layers)
    store = storage_bf16(need_dx0)
    pre, fdesc, counters, cout, poffs, ptotal, n_out, cb, start, total, nbytes = _level_desc(encoder, layers, params, store)
    arena = torch.empty(total, dtype=torch.uint8, device=dev)
    stats = torch.empty((nl, 2, 128), dtype=_f32, device=dev)       # mean / invstd rows (cout <= 128)
    base, sbase = arena.data_ptr(), stats.data_ptr()
    desc = pre.copy()                        # (C / Y / MEAN / INVSTD hold offsets: add this pass's arena and stats addresses)
    prof = _profile_slots(layers, store) if F_.PROFILE is not None else None
    if prof is not None:
        desc[:, _E["PROF"]] = prof[1]
    desc[:, _E["C"]] += base
    desc[:, _E["Y"]] += base
    desc[0, _E["X"]] = x0.data_ptr()
    desc[1:, _E["X"]] = desc[:-1, _E["Y"]]
    desc[:, _E["MEAN"]] += sbase
    desc[:, _E["INVSTD"]] += sbase
    ws = _ws(nbytes, dev)
    lane = lane_of(encoder)
    group = _sync_group_for(layers)
    st = Launched()
    st.sync = None
    if group is not None:
        lane = None
        sums = torch.zeros(nl * SYNC_STRIDE, dtype=torch.float64, device=dev)
        key = id(sums)
        _SYNC_CTX[key] = (sums, None, group)
        try:
            rc = lib.irx_encoder_forward_sync(desc.ctypes.data, fdesc.ctypes.data, nl, ws.data_ptr(), nbytes,
                                              _lib.stream_ptr(), sums.data_ptr(), _ALLREDUCE_C, key)
        finally:
            del _SYNC_CTX[key]
        if rc and _SYNC_ERR:
            raise RuntimeError("sync BatchNorm all-reduce inside the encoder executor failed: %s" % _SYNC_ERR.pop())
        st.sync = (sums, group)
        SYNC_CALLS[0] += 1
    elif lane is not None:
        rc = lib.irx_encoder_submit(lane, 0, desc.ctypes.data, fdesc.ctypes.data, nl, None, None, ws.data_ptr(),
                                    nbytes, _lib.stream_ptr())
        _HELD.setdefault(lane, []).append((ws, x0, arena, stats, prof))
    else:
        rc = lib.irx_encoder_forward(desc.ctypes.data, fdesc.ctypes.data, nl, ws.data_ptr(), nbytes, _lib.stream_ptr())
    if rc:
        _lib.check(rc, "irx_encoder_forward")
    st.lane = lane
    if counters:
        from .. import _counters
        _counters.bump(counters)             # (one launch per forward when InstanceRefer collects them: _counters.py)
    st.layers, st.desc, st.fdesc, st.extra = layers, desc, fdesc, (n_out, cout, poffs, ptotal)
    st.store, st.prof, st.need_dx0 = store, prof, bool(need_dx0)
    st.level0 = layers.root()            # the plan only holds its finest level weakly (Plan): the pass and its node own it
    if st.level0 is None:
        raise RuntimeError("encoder pass launched over a coordinate pyramid that no longer exists")
    if TRACE is not None:
        TRACE["fwd"] = dict(layers=layers, arena=arena, stats=stats, x0=x0, start=start, cb=cb, n_out=n_out, cout=cout,
                            store=store)
    # gradient sink (optim.FlatAdam): parameter gradients can go straight into the optimizer's flat buffer
    sink = getattr(params[0], "_irx_sink", None)
    st.sink = (sink[0], id(encoder), params) if sink is not None else None
    st.saved = (x0, arena, stats)
    o0 = int(start[-1] + cb[-1])
    st.out = arena[o0:o0 + 4 * layers[-1].n_out * layers[-1].cout].view(_f32).view(layers[-1].n_out, layers[-1].cout)
    return st


class EncoderFn(torch.autograd.Function):
    """forward / backward = one irx_encoder_forward / irx_encoder_backward call over a descriptor table; activations,
    gradients-in-flight and parameter gradients live in three arenas allocated once per call. The table is a cached
    static template plus a handful of vectorised numpy fills (level sizes, table pointers, arena offsets).
    apply(feats, encoder, layers | Launched, *params): with a Launched pass (launch() above) the forward only binds it."""

    @staticmethod
    def forward(ctx, feats, encoder, layers, *params):
        # compact form (apply_encoder below): only the FIRST parameter is an autograd input, the others ride in a _Hidden — their
        # gradients go to the optimizer's sink, and 38 AccumulateGrad nodes that would each be evaluated with an undefined gradient
        # (plus 38 saved variables) per encoder pass are not created
        ctx.n_inputs = 3 + len(params)
        if params and isinstance(params[-1], _Hidden):
            params = params[-1].params
        ctx.all_params = params
        st = layers if isinstance(layers, Launched) else launch(feats, encoder, layers, params, ctx.needs_input_grad[0])
        if ctx.needs_input_grad[0] and not st.need_dx0:
            raise RuntimeError("encoder pass was launched without an input gradient (bf16 storage) but its features require one")
        ctx.lane, ctx.sync = st.lane, st.sync
        ctx.layers, ctx.desc, ctx.fdesc, ctx.extra = st.layers, st.desc, st.fdesc, st.extra
        ctx.store, ctx.prof, ctx.sink, ctx.level0 = st.store, st.prof, st.sink, st.level0
        ctx.gate = getattr(encoder, '_irx_bwd_gate', None)   # (role, rows, token): irx_encoder_gate_next, set per step by the model
        ctx.wstream = getattr(encoder, '_irx_wgrad_stream', None)   # hipStream_t (int) lent for the backward's weight gradients, or None
        ctx.save_for_backward(*st.saved)
        return st.out.view_as(st.out) if isinstance(layers, Launched) else st.out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        layers, nl = ctx.layers, len(ctx.layers)
        n_out, cout, poffs, ptotal = ctx.extra
        dev = dout.device
        dout = dout.contiguous().float()
        # gradients in flight (bytes): gy_i for every layer but the last (that one IS dout) + the shared dc scratch
        store = ctx.store                # (the table carries the forward pass's compute mode: IRX_ENC_MODE)
        gsz = _up256(n_out * cout * (2 if store else 4))
        goffs = np.concatenate([[0], np.cumsum(gsz[:-1])])              # nl entries; the last one = start of dc
        dc_off = int(goffs[-1])
        dc_size = int(gsz.max())
        two = WGRAD_STREAM and ctx.sync is None and ctx.wstream is not None
        total = dc_off + dc_size * (2 if two else 1)
        garena = torch.empty(total, dtype=torch.uint8, device=dev)
        slots = None
        if ctx.sink is not None:
            owner, key, sparams = ctx.sink
            slots = owner.sink_slots(key, sparams)
        if slots is None:
            pgrad = torch.empty(ptotal, dtype=_f32, device=dev)        # kernel, gamma, beta gradients of every layer
            pptr = pgrad.data_ptr() + poffs
        else:
            pptr = np.fromiter((t.data_ptr() for t in slots), dtype=np.int64, count=3 * nl).reshape(nl, 3)
        gbase = garena.data_ptr()
        if TRACE is not None:
            TRACE["bwd"] = dict(garena=garena, goffs=goffs, dout=dout, store=store)
        desc = ctx.desc.copy()
        need_dx0 = ctx.needs_input_grad[0]
        ev = layers.pre.pop("bwd_event", None) if isinstance(layers, Plan) else None
        if ev is not None:                                  # tables prebuilt on another stream (prebuild_backward)
            torch.cuda.current_stream().wait_event(ev)
        _backward_tables(layers)                            # pair lists / transposed maps still missing: built here
        tb, pr = [], []
        for L in layers:
            if L.down:
                tbl_b, ld_b = L.lv_in.down().child_t()
                tb.append((tbl_b.data_ptr(), ld_b, 0))
            else:
                tb.append((L.tbl.data_ptr(), L.ld, 1))
            if _uses_pairs(L):
                il, ol, counts, ldp = L.lv_in.down().pairs() if L.down else L.lv_in.pairs27()
                pr.append((il.data_ptr(), ol.data_ptr(), counts.data_ptr(), ldp))
            else:
                pr.append((0, 0, 0, 0))
        desc[:, _E["TBL_B"]:_E["FLIP_B"] + 1] = np.array(tb, dtype=np.int64)
        desc[:, _E["PAIR_IN"]:_E["LD_PAIRS"] + 1] = np.array(pr, dtype=np.int64)
        desc[:, _E["DW"]:_E["DBETA"] + 1] = pptr
        desc[:-1, _E["GY"]] = gbase + goffs[:-1]
        desc[-1, _E["GY"]] = dout.data_ptr()
        if two:                          # weight gradients on a second stream the model lends (irx.h IRX_ENC_DC2 / IRX_ENC_WSTREAM)
            desc[0, _E["DC2"]] = gbase + dc_off + dc_size
            desc[0, _E["WSTREAM"]] = int(ctx.wstream)
            desc[0, _E["WROWS"]] = WGRAD_STREAM_ROWS
        dfeats = torch.empty((layers[0].n_in, layers[0].cin), dtype=_f32, device=dev) if need_dx0 else None
        fdesc = ctx.fdesc
        nbytes = lib.irx_encoder_workspace_bytes(desc.ctypes.data, fdesc.ctypes.data, nl, 1)
        ws = _ws(nbytes, dev)
        if ctx.sync is not None:
            sums, group = ctx.sync
            gsums = torch.empty(nl * SYNC_STRIDE, dtype=_f32, device=dev)
            key = id(gsums)
            _SYNC_CTX[key] = (sums, gsums, group)
            try:
                rc = lib.irx_encoder_backward_sync(desc.ctypes.data, fdesc.ctypes.data, nl, gbase + dc_off,
                                                   dfeats.data_ptr() if need_dx0 else None, ws.data_ptr(), nbytes,
                                                   _lib.stream_ptr(), sums.data_ptr(), gsums.data_ptr(), _ALLREDUCE_C, key)
            finally:
                del _SYNC_CTX[key]
            if rc:
                if _SYNC_ERR:
                    raise RuntimeError("sync BatchNorm all-reduce inside the encoder executor failed: %s" % _SYNC_ERR.pop())
                _lib.check(rc, "irx_encoder_backward_sync")
            SYNC_CALLS[1] += 1
            if slots is not None:
                owner.sink_delivered(ctx.sink[1], sparams)
                return _returns(ctx, dfeats, None)
            return _returns(ctx, dfeats, _grad_views(pgrad, poffs, layers))
        if ctx.lane is not None and slots is not None and not need_dx0:
            # nothing autograd will touch depends on this pass: a library thread issues it; the optimizer waits for the
            # lane before it records the delivery event (FlatAdam.gather_grads)
            if ctx.gate is not None:
                lib.irx_encoder_gate_next(*ctx.gate)
            rc = lib.irx_encoder_submit(ctx.lane, 1, desc.ctypes.data, fdesc.ctypes.data, nl, gbase + dc_off, None,
                                        ws.data_ptr(), nbytes, _lib.stream_ptr())
            if rc:
                _lib.check(rc, "irx_encoder_submit")
            _HELD.setdefault(ctx.lane, []).append((ws, garena, dout, slots, ctx.saved_tensors, ctx.prof))
            owner.sink_delivered(key, sparams, lane=ctx.lane)
            return _returns(ctx, None, None)
        lane_wait(ctx.lane)                                  # same-stream order with the (possibly queued) forward
        if ctx.gate is not None:
            lib.irx_encoder_gate_next(*ctx.gate)
        rc = lib.irx_encoder_backward(desc.ctypes.data, fdesc.ctypes.data, nl, gbase + dc_off,
                                      dfeats.data_ptr() if need_dx0 else None, ws.data_ptr(), nbytes,
                                      _lib.stream_ptr())
        if rc:
            _lib.check(rc, "irx_encoder_backward")
        if slots is not None:
            owner.sink_delivered(key, sparams)
            return _returns(ctx, dfeats, None)
        return _returns(ctx, dfeats, _grad_views(pgrad, poffs, layers))


class _Hidden:
    """Parameters handed to EncoderFn without being autograd inputs (see EncoderFn.forward)."""
    __slots__ = ("params",)

    def __init__(self, params):
        self.params = tuple(params)


def apply_encoder(feats, encoder, layers, params):
    """EncoderFn.apply in its compact form when the parameters' gradients can go to an optimizer's sink (optim.FlatAdam re-homed
    them), else with every parameter as an input."""
    # (the compact form's only autograd input is params[0] — a frozen stem kernel, or any frozen parameter, would leave
    #  the whole encoder without a node or some parameters without a gradient path: every parameter must require grad)
    if getattr(params[0], "_irx_sink", None) is not None and not feats.requires_grad and all(p.requires_grad for p in params):
        return EncoderFn.apply(feats, encoder, layers, params[0], _Hidden(params))
    return EncoderFn.apply(feats, encoder, layers, *params)


def _grad_views(pgrad, poffs, layers):
    grads = []
    po = poffs // 4
    for i, L in enumerate(layers):
        o = po[i]
        grads.append(pgrad[o[0]:o[0] + L.K * L.cin * L.cout].view(L.K, L.cin, L.cout))
        grads.append(pgrad[o[1]:o[1] + L.cout])
        grads.append(pgrad[o[2]:o[2] + L.cout])
    return grads


def _returns(ctx, dfeats, grads):
    """backward()'s return tuple. grads None: delivered through the sink. Compact form with ordinary gradients (a second backward
    before zero_grad(), a sink that refused): the hidden parameters' gradients are accumulated into .grad by hand, exactly what their
    AccumulateGrad nodes would have done."""
    n_params = ctx.n_inputs - 3
    if grads is None:
        return (dfeats, None, None) + (None,) * n_params
    if n_params == len(grads):
        return (dfeats, None, None) + tuple(grads)
    with torch.no_grad():
        for p, g in zip(ctx.all_params[1:], grads[1:]):
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.add_(g)
    return (dfeats, None, None, grads[0], None)


PROFILE_NO_COUNT = False     # timeline mode: only the events matter; no pair counting (it syncs)
EVENT_POOL = []        # instantiated timing events (bench.py's timeline mode fills it ahead of the loop: creating 156 events per
                       # step inside the loop costs the host ~2 ms and turns the loop host-paced)


def _new_event():
    e = torch.cuda.Event(enable_timing=True)
    e.record()                                            # instantiates the HIP event; re-recorded by the library
    return e


def _profile_slots(layers, store):
    """bench.py's instrumented steps: six timing events per layer (start / stop around the dominant forward, data-gradient
    and weight-gradient kernel; recorded by the library on the launch stream: IRX_ENC_PROF) and the matching PROFILE
    records (kind, n_out, K, cin, cout, pairs, start, stop, element size of the gathered / written activations)."""
    import ctypes
    nl = len(layers)
    evs, handles = [], (ctypes.c_void_p * (6 * nl))()
    for i, L in enumerate(layers):
        row = [EVENT_POOL.pop() if EVENT_POOL else _new_event() for _ in range(6)]
        for j, e in enumerate(row):
            handles[6 * i + j] = e.cuda_event
        evs.append(row)
        m = 0 if PROFILE_NO_COUNT else F_._pairs(L.tbl, L.K, L.n_out)      # (a host sync per fresh table)
        esz = 2 if (store and i > 0) else 4
        F_.PROFILE.append(("fwd", L.n_out, L.K, L.cin, L.cout, m, row[0], row[1], esz))
        if i > 0:
            F_.PROFILE.append(("dgrad", L.n_in, L.K, L.cout, L.cin, m, row[2], row[3], 2 if store else 4))
        F_.PROFILE.append(("wgrad", L.n_out, L.K, L.cin, L.cout, m, row[4], row[5], 2 if store else 4))
    base = ctypes.addressof(handles)
    ptrs = base + 6 * ctypes.sizeof(ctypes.c_void_p) * np.arange(nl, dtype=np.int64)
    return (evs, handles), ptrs


def run_encoder(encoder, st, defer=False):
    """Fused training forward of a SparseConvEncoder on a canonical SparseTensor -> SparseTensor at stride 16.
    defer=True: issue the pass now, create its autograd node later -> Deferred (see Launched); .attach() -> SparseTensor."""
    from .tensor import SparseTensor
    layers = build_plan(encoder, st.level())
    params = []
    for L in layers:
        params += [L.conv.kernel, L.bn.weight, L.bn.bias]
    out = layers[-1].lv_out
    if defer and not st.F.requires_grad:
        return Deferred(st.F, encoder, launch(st.F, encoder, layers, params, False), params, out)
    y = apply_encoder(st.F, encoder, layers, params)
    return SparseTensor(y, out.coords, out.stride, out.batch_size, out)


class Deferred:
    """An issued encoder pass waiting for its autograd node (run_encoder(defer=True))."""
    __slots__ = ("feats", "encoder", "launched", "params", "level", "stream")

    def __init__(self, feats, encoder, launched, params, level):
        self.feats, self.encoder, self.launched, self.params, self.level = feats, encoder, launched, params, level
        # autograd replays a node on the stream that was current when the node was CREATED: attach() restores the issue stream
        self.stream = torch.cuda.current_stream(feats.device) if feats.is_cuda else None

    def record_stream(self, stream):
        for t in self.launched.saved:
            t.record_stream(stream)

    def attach(self):
        from .tensor import SparseTensor
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                y = apply_encoder(self.feats, self.encoder, self.launched, self.params)
        else:
            y = apply_encoder(self.feats, self.encoder, self.launched, self.params)
        out = self.level
        return SparseTensor(y, out.coords, out.stride, out.batch_size, out)


def can_fuse(encoder):
    """The executor covers the training configuration of the reference (train-mode BatchNorm, fp32, no bias)."""
    if not (encoder.training and torch.is_grad_enabled()):
        return False
    if not all(bn.training for _, bn, _, _ in _skeleton(encoder)):
        return False                                   # frozen BatchNorm layers: the per-layer path handles eval statistics
    if not SYNC_IN_EXECUTOR and any(getattr(bn, "_irx_sync", False) for _, bn, _, _ in _skeleton(encoder)) \
            and F_.sync_group() is not None:
        return False                                   # sync BatchNorm layer by layer (IRX_SYNC_BN_EXECUTOR=0: the round-2/3 path)
    ok = encoder.__dict__.get("_irx_fusable")          # structural part: decided once per encoder instance
    if ok is None:
        ok = True
        for m in encoder.modules():
            if isinstance(m, torch.nn.BatchNorm1d) and (not m.track_running_stats or m.weight is None or m.momentum is None):
                ok = False
            if hasattr(m, "kernel") and getattr(m, "bias", None) is not None:
                ok = False
        encoder.__dict__["_irx_fusable"] = ok
    return ok
