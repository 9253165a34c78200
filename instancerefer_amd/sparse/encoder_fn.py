"""Whole-encoder executor: the 13 Conv3d->BatchNorm(->+res)->ReLU groups of SparseConvEncoder / BEVEncoder
(reference models/basic_blocks.py:59-95,136-171) issued as ONE autograd node.

The per-layer path (sparse/nn.py: conv_bn_act) costs ~3 autograd nodes, ~10 tensor allocations and ~100 us of Python
per layer and direction; with two encoders that was ~10 ms of host time per step, i.e. the step was host-bound.
Here forward and backward are tight loops of C-ABI calls (the same kernels, in the same order, so results are
bit-identical to the per-layer path) with shared workspaces; Python touches each layer once.

Layer list (index: conv, residual source): 0 stem | per stage s: 3s+1 down (2^3/2), 3s+2 res-a, 3s+3 res-b (+ out of 3s+1).
"""
import torch

from .. import _lib
from . import functional as F_

_f32 = torch.float32
_PAIR = (32, 64, 128)


class _Layer:
    __slots__ = ("conv", "bn", "lv_in", "lv_out", "K", "cin", "cout", "tbl", "ld", "n_in", "n_out", "down", "res")


def build_plan(encoder, level0):
    """Resolve the coordinate levels / tables of every layer (all cached on the levels; the pyramid is already built)."""
    layers = []

    def add(block_conv, block_bn, lv_in, down, res):
        L = _Layer()
        L.conv, L.bn, L.lv_in, L.down, L.res = block_conv, block_bn, lv_in, down, res
        L.K, L.cin, L.cout = block_conv.kernel.shape
        if down:
            dm = lv_in.down()
            L.lv_out, L.tbl, L.ld = dm.out_level, dm.child, dm.ld
        else:
            L.lv_out = lv_in
            L.tbl, L.ld = lv_in.nbr27()
        L.n_in, L.n_out = lv_in.n, L.lv_out.n
        layers.append(L)
        return L.lv_out

    lv = add(encoder.stem[0].net[0], encoder.stem[0].net[1], level0, False, -1)
    for stage in (encoder.stage1, encoder.stage2, encoder.stage3, encoder.stage4):
        lv = add(stage[0].net[0], stage[0].net[1], lv, True, -1)
        d = len(layers) - 1
        rb = stage[1]
        if len(rb.downsample) != 0:
            raise NotImplementedError("encoder executor: projection shortcuts are not part of the InstanceRefer encoders")
        lv = add(rb.net[0], rb.net[1], lv, False, -1)
        lv = add(rb.net[3], rb.net[4], lv, False, d)
    return layers


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, layers, *params):
        lib = _lib.load()
        stream = _lib.stream_ptr()
        dev = feats.device
        x = feats.contiguous().float()
        nl = len(layers)
        # shared workspaces (stream-ordered reuse)
        conv_ws = max(lib.irx_spconv_fwd_workspace_bytes(L.n_out, L.K, L.cin, L.cout, 0) for L in layers)
        bn_ws = max(lib.irx_bn_workspace_bytes(L.n_out, L.cout) for L in layers)
        ws_c, ws_b = _ws(conv_ws, dev), _ws(bn_ws, dev)
        pc, pb = ws_c.data_ptr(), ws_b.data_ptr()
        stats = torch.empty((nl, 2, 128), dtype=_f32, device=dev)       # mean / invstd rows (cout <= 128)
        xs, cs, ys = [], [], []
        for i, L in enumerate(layers):
            w, gamma, beta = params[3 * i], params[3 * i + 1], params[3 * i + 2]
            c = torch.empty((L.n_out, L.cout), dtype=_f32, device=dev)
            y = torch.empty((L.n_out, L.cout), dtype=_f32, device=dev)
            rc = lib.irx_spconv_fwd(x.data_ptr(), w.data_ptr(), L.tbl.data_ptr(), L.ld, L.n_out, L.K, L.cin, L.cout,
                                    0, 0, c.data_ptr(), pc, conv_ws, stream)
            if rc:
                _lib.check(rc, "irx_spconv_fwd")
            bn = L.bn
            mean_p, inv_p = stats[i, 0].data_ptr(), stats[i, 1].data_ptr()
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            rc = lib.irx_bn_stats(c.data_ptr(), L.n_out, L.cout, bn.eps, 0.0 if bn.momentum is None else bn.momentum,
                                  mean_p, inv_p, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), pb, bn_ws, stream)
            if rc:
                _lib.check(rc, "irx_bn_stats")
            res_p = ys[L.res].data_ptr() if L.res >= 0 else None
            rc = lib.irx_bn_apply(c.data_ptr(), L.n_out, L.cout, mean_p, inv_p, gamma.data_ptr(), beta.data_ptr(),
                                  res_p, 1, y.data_ptr(), stream)
            if rc:
                _lib.check(rc, "irx_bn_apply")
            xs.append(x)
            cs.append(c)
            ys.append(y)
            x = y
        ctx.layers = layers
        ctx.save_for_backward(stats, *xs, *cs, *ys, *params)
        return x

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        stream = _lib.stream_ptr()
        layers = ctx.layers
        nl = len(layers)
        saved = ctx.saved_tensors
        stats = saved[0]
        xs, cs, ys = saved[1:1 + nl], saved[1 + nl:1 + 2 * nl], saved[1 + 2 * nl:1 + 3 * nl]
        params = saved[1 + 3 * nl:]
        dev = dout.device
        conv_ws = max(lib.irx_spconv_fwd_workspace_bytes(L.n_in, L.K, L.cout, L.cin, 1) for L in layers)
        wg_ws = 0
        for L in layers:
            if L.cin in _PAIR and L.cout in _PAIR:
                wg_ws = max(wg_ws, lib.irx_spconv_wgrad_pairs_workspace_bytes(L.n_out, L.K, L.cin, L.cout))
            else:
                wg_ws = max(wg_ws, lib.irx_spconv_wgrad_workspace_bytes(L.n_out, L.K, L.cin, L.cout))
        bn_ws = max(lib.irx_bn_workspace_bytes(L.n_out, L.cout) for L in layers)
        ws_c, ws_w, ws_b = _ws(conv_ws, dev), _ws(wg_ws, dev), _ws(bn_ws, dev)
        pc, pw, pb = ws_c.data_ptr(), ws_w.data_ptr(), ws_b.data_ptr()
        grads = [None] * (3 * nl)
        gy = [None] * nl                      # gradient w.r.t. each layer's output
        gy[nl - 1] = dout.contiguous().float()
        dfeats = None
        for i in range(nl - 1, -1, -1):
            L = layers[i]
            w, gamma = params[3 * i], params[3 * i + 1]
            g = gy[i]
            dc = torch.empty((L.n_out, L.cout), dtype=_f32, device=dev)
            dgamma = torch.empty(L.cout, dtype=_f32, device=dev)
            dbeta = torch.empty(L.cout, dtype=_f32, device=dev)
            dres = torch.empty((L.n_out, L.cout), dtype=_f32, device=dev) if L.res >= 0 else None
            rc = lib.irx_bn_backward(cs[i].data_ptr(), ys[i].data_ptr(), g.data_ptr(), L.n_out, L.cout,
                                     stats[i, 0].data_ptr(), stats[i, 1].data_ptr(), gamma.data_ptr(), 1,
                                     dc.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                     dres.data_ptr() if dres is not None else None, pb, bn_ws, stream)
            if rc:
                _lib.check(rc, "irx_bn_backward")
            if dres is not None:
                gy[L.res] = dres              # the shortcut's share; the main path is added below when it arrives
            # weight gradient
            dw = torch.empty((L.K, L.cin, L.cout), dtype=_f32, device=dev)
            if L.cin in _PAIR and L.cout in _PAIR:
                il, ol, counts, ldp = L.lv_in.down().pairs() if L.down else L.lv_in.pairs27()
                rc = lib.irx_spconv_wgrad_pairs(xs[i].data_ptr(), dc.data_ptr(), il.data_ptr(), ol.data_ptr(), ldp,
                                                counts.data_ptr(), L.n_out, L.K, L.cin, L.cout, dw.data_ptr(), pw, wg_ws,
                                                stream)
            else:
                rc = lib.irx_spconv_wgrad(xs[i].data_ptr(), dc.data_ptr(), L.tbl.data_ptr(), L.ld, L.n_out, L.K, L.cin,
                                          L.cout, dw.data_ptr(), pw, wg_ws, stream)
            if rc:
                _lib.check(rc, "irx_spconv_wgrad")
            grads[3 * i], grads[3 * i + 1], grads[3 * i + 2] = dw, dgamma, dbeta
            # data gradient
            if i > 0 or ctx.needs_input_grad[0]:
                if L.down:
                    tbl_b, ld_b = L.lv_in.down().child_t()
                    flip = 0
                else:
                    tbl_b, ld_b, flip = L.tbl, L.ld, 1
                dx = torch.empty((L.n_in, L.cin), dtype=_f32, device=dev)
                rc = lib.irx_spconv_fwd(dc.data_ptr(), w.data_ptr(), tbl_b.data_ptr(), ld_b, L.n_in, L.K, L.cout, L.cin,
                                        flip, 1, dx.data_ptr(), pc, conv_ws, stream)
                if rc:
                    _lib.check(rc, "irx_spconv_fwd(dgrad)")
                if i == 0:
                    dfeats = dx
                elif gy[i - 1] is None:
                    gy[i - 1] = dx
                else:
                    gy[i - 1] = gy[i - 1].add_(dx)   # shortcut share (dres) + main path
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
    for m in encoder.modules():
        if isinstance(m, torch.nn.BatchNorm1d) and (not m.track_running_stats or m.weight is None):
            return False
        if hasattr(m, "kernel") and getattr(m, "bias", None) is not None:
            return False
    return True
