"""Whole-encoder executor: the 13 Conv3d->BatchNorm(->+res)->ReLU groups of SparseConvEncoder / BEVEncoder
(reference models/basic_blocks.py:59-95,136-171) issued as ONE autograd node.

The per-layer path (sparse/nn.py: conv_bn_act) costs ~3 autograd nodes, ~10 tensor allocations and ~100 us of Python
per layer and direction; with two encoders that was ~10 ms of host time per step, i.e. the step was host-bound.
Here forward and backward are ONE C-ABI call each (irx_encoder_forward / irx_encoder_backward, include/irx.h): the
library walks a descriptor table and launches the same kernels in the same order as the per-layer path (bit-identical
results); Python only fills the table and allocates three arenas.

Layer list (index: conv, residual source): 0 stem | per stage s: 3s+1 down (2^3/2), 3s+2 res-a, 3s+3 res-b (+ out of 3s+1).
"""
import numpy as np
import torch

from .. import _lib
from . import functional as F_

_f32 = torch.float32
_PAIR = (32, 64, 128)


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


def build_plan(encoder, level0):
    """Bind the coordinate levels / tables of this batch to the layer skeleton (the pyramid is already built)."""
    layers = []
    lv = level0
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
    return layers


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


# field order of the descriptor table: include/irx.h, enum IRX_ENC_* (tests/test_abi_cpu.py checks the two agree)
ENC_FIELDS = ("K", "CIN", "COUT", "N_IN", "N_OUT", "RES", "TBL", "LD", "TBL_B", "LD_B", "FLIP_B", "PAIR_IN", "PAIR_OUT",
              "PAIR_COUNTS", "LD_PAIRS", "W", "GAMMA", "BETA", "RUNNING_MEAN", "RUNNING_VAR", "X", "C", "Y", "MEAN",
              "INVSTD", "DW", "DGAMMA", "DBETA", "GY")
_E = {n: i for i, n in enumerate(ENC_FIELDS)}
_NF = len(ENC_FIELDS)
_ALIGN = 64                                           # float32 elements (256 B)


def _up(n):
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


class EncoderFn(torch.autograd.Function):
    """forward / backward = one irx_encoder_forward / irx_encoder_backward call over a descriptor table; activations,
    gradients-in-flight and parameter gradients live in three arenas allocated once per call."""

    @staticmethod
    def forward(ctx, feats, layers, *params):
        lib = _lib.load()
        dev = feats.device
        x0 = feats.contiguous().float()
        nl = len(layers)
        # activation arena: conv output c_i and layer output y_i of every layer
        offs, total = [], 0
        for L in layers:
            n = _up(L.n_out * L.cout)
            offs.append((total, total + n))
            total += 2 * n
        arena = torch.empty(total, dtype=_f32, device=dev)
        stats = torch.empty((nl, 2, 128), dtype=_f32, device=dev)       # mean / invstd rows (cout <= 128)
        base, sbase = arena.data_ptr(), stats.data_ptr()
        rows, frows, counters = [], [], []
        for i, L in enumerate(layers):
            bn = L.bn
            r = [0] * _NF
            r[_E["K"]], r[_E["CIN"]], r[_E["COUT"]] = L.K, L.cin, L.cout
            r[_E["N_IN"]], r[_E["N_OUT"]], r[_E["RES"]] = L.n_in, L.n_out, L.res
            r[_E["TBL"]], r[_E["LD"]] = L.tbl.data_ptr(), L.ld
            r[_E["W"]] = params[3 * i].data_ptr()
            r[_E["GAMMA"]], r[_E["BETA"]] = params[3 * i + 1].data_ptr(), params[3 * i + 2].data_ptr()
            r[_E["RUNNING_MEAN"]], r[_E["RUNNING_VAR"]] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            r[_E["C"]], r[_E["Y"]] = base + 4 * offs[i][0], base + 4 * offs[i][1]
            r[_E["X"]] = x0.data_ptr() if i == 0 else rows[i - 1][_E["Y"]]
            r[_E["MEAN"]], r[_E["INVSTD"]] = sbase + 4 * (i * 256), sbase + 4 * (i * 256 + 128)
            rows.append(r)
            frows.append((bn.eps, 0.0 if bn.momentum is None else bn.momentum))
            if bn.num_batches_tracked is not None:
                counters.append(bn.num_batches_tracked)
        desc = np.array(rows, dtype=np.int64)
        fdesc = np.array(frows, dtype=np.float64)
        nbytes = lib.irx_encoder_workspace_bytes(desc.ctypes.data, fdesc.ctypes.data, nl, 0)
        ws = _ws(nbytes, dev)
        rc = lib.irx_encoder_forward(desc.ctypes.data, fdesc.ctypes.data, nl, ws.data_ptr(), nbytes, _lib.stream_ptr())
        if rc:
            _lib.check(rc, "irx_encoder_forward")
        if counters:
            torch._foreach_add_(counters, 1)
        ctx.layers, ctx.desc, ctx.fdesc = layers, desc, fdesc
        ctx.save_for_backward(x0, arena, stats, *params)
        o0 = offs[-1][1]
        return arena[o0:o0 + layers[-1].n_out * layers[-1].cout].view(layers[-1].n_out, layers[-1].cout)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        layers, nl = ctx.layers, len(ctx.layers)
        x0, arena, stats = ctx.saved_tensors[:3]
        dev = dout.device
        dout = dout.contiguous().float()
        # gradients in flight: gy_i for every layer but the last (that one IS dout) + the shared dc scratch
        goffs, total, dc_max = [], 0, 0
        for L in layers[:-1]:
            goffs.append(total)
            total += _up(L.n_out * L.cout)
        for L in layers:
            dc_max = max(dc_max, L.n_out * L.cout)
        dc_off = total
        total += _up(dc_max)
        garena = torch.empty(total, dtype=_f32, device=dev)
        # parameter gradients, in parameter order (kernel, gamma, beta per layer)
        poffs, ptotal = [], 0
        for L in layers:
            o = [ptotal]
            ptotal += _up(L.K * L.cin * L.cout)
            o.append(ptotal)
            ptotal += _up(L.cout)
            o.append(ptotal)
            ptotal += _up(L.cout)
            poffs.append(o)
        pgrad = torch.empty(ptotal, dtype=_f32, device=dev)
        gbase, pbase = garena.data_ptr(), pgrad.data_ptr()
        desc = ctx.desc.copy()
        need_dx0 = ctx.needs_input_grad[0]
        for i, L in enumerate(layers):
            r = desc[i]
            if L.down:
                tbl_b, ld_b = L.lv_in.down().child_t()
                r[_E["TBL_B"]], r[_E["LD_B"]], r[_E["FLIP_B"]] = tbl_b.data_ptr(), ld_b, 0
            else:
                r[_E["TBL_B"]], r[_E["LD_B"]], r[_E["FLIP_B"]] = L.tbl.data_ptr(), L.ld, 1
            if L.cin in _PAIR and L.cout in _PAIR:
                il, ol, counts, ldp = L.lv_in.down().pairs() if L.down else L.lv_in.pairs27()
                r[_E["PAIR_IN"]], r[_E["PAIR_OUT"]] = il.data_ptr(), ol.data_ptr()
                r[_E["PAIR_COUNTS"]], r[_E["LD_PAIRS"]] = counts.data_ptr(), ldp
            r[_E["DW"]], r[_E["DGAMMA"]], r[_E["DBETA"]] = (pbase + 4 * poffs[i][0], pbase + 4 * poffs[i][1],
                                                          pbase + 4 * poffs[i][2])
            r[_E["GY"]] = gbase + 4 * goffs[i] if i < nl - 1 else dout.data_ptr()
        dfeats = torch.empty((layers[0].n_in, layers[0].cin), dtype=_f32, device=dev) if need_dx0 else None
        fdesc = ctx.fdesc
        nbytes = lib.irx_encoder_workspace_bytes(desc.ctypes.data, fdesc.ctypes.data, nl, 1)
        ws = _ws(nbytes, dev)
        rc = lib.irx_encoder_backward(desc.ctypes.data, fdesc.ctypes.data, nl, gbase + 4 * dc_off,
                                      dfeats.data_ptr() if need_dx0 else None, ws.data_ptr(), nbytes,
                                      _lib.stream_ptr())
        if rc:
            _lib.check(rc, "irx_encoder_backward")
        grads = []
        for i, L in enumerate(layers):
            o = poffs[i]
            grads.append(pgrad[o[0]:o[0] + L.K * L.cin * L.cout].view(L.K, L.cin, L.cout))
            grads.append(pgrad[o[1]:o[1] + L.cout])
            grads.append(pgrad[o[2]:o[2] + L.cout])
        return (dfeats, None) + tuple(grads)


def run_encoder(encoder, st):
    """Fused training forward of a SparseConvEncoder on a canonical SparseTensor -> SparseTensor at stride 16."""
    from .tensor import SparseTensor
    layers = build_plan(encoder, st.level())
    params = []
    for L in layers:
        params += [L.conv.kernel, L.bn.weight, L.bn.bias]
    y = EncoderFn.apply(st.F, layers, *params)
    out = layers[-1].lv_out
    return SparseTensor(y, out.coords, out.stride, out.batch_size, out)


def can_fuse(encoder):
    """The executor covers the training configuration of the reference (train-mode BatchNorm, fp32, no bias)."""
    if not (encoder.training and torch.is_grad_enabled()) or F_.PROFILE is not None:
        return False
    ok = encoder.__dict__.get("_irx_fusable")          # structural part: decided once per encoder instance
    if ok is None:
        ok = True
        for m in encoder.modules():
            if isinstance(m, torch.nn.BatchNorm1d) and (not m.track_running_stats or m.weight is None):
                ok = False
            if hasattr(m, "kernel") and getattr(m, "bias", None) is not None:
                ok = False
        encoder.__dict__["_irx_fusable"] = ok
    return ok
