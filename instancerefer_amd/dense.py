"""Dense sequence ops of the language encoder on the irx kernels: a persistent GRU recurrence
(csrc/irx_gru.hip) with the time-parallel projections as GEMMs; plus the fused matching-score / contrastive-loss ops
(csrc/irx_match.hip).

`gru_packed(gru, x, lengths)` reproduces `pad_packed_sequence(gru(pack_padded_sequence(x, lengths)))` of an
`nn.GRU(batch_first=True)` (reference models/lang_module.py:53-57) using that module's own parameters
(state-dict keys unchanged)."""
import math
import os

import torch

from . import _lib

_f32 = torch.float32


class GRULayerFn(torch.autograd.Function):
    """One GRU layer, both directions. x (B,T,I); w_ih (ndir*3H, I); b_ih (ndir*3H); w_hh (ndir,3H,H); b_hh (ndir,3H)."""

    @staticmethod
    def forward(ctx, x, lengths_i32, w_ih, b_ih, w_hh, b_hh):
        B, T, I = x.shape
        ndir, threeH, H = w_hh.shape
        x2 = x.reshape(B * T, I)
        gi = torch.addmm(b_ih, x2, w_ih.t()).contiguous()                  # (B*T, ndir*3H)
        out = torch.empty((B, T, ndir * H), dtype=_f32, device=x.device)
        gates = torch.empty((B, T, ndir, 4 * H), dtype=_f32, device=x.device)
        w_hh_c = w_hh.contiguous()
        b_hh_c = b_hh.contiguous()
        _lib.call("irx_gru_forward", _lib.ptr(gi), _lib.ptr(lengths_i32), _lib.ptr(w_hh_c), _lib.ptr(b_hh_c),
                  B, T, ndir, H, _lib.ptr(out), _lib.ptr(gates), _lib.stream_ptr())
        ctx.save_for_backward(x2, lengths_i32, w_ih, w_hh_c, out, gates)
        ctx.dims = (B, T, I, ndir, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, lengths_i32, w_ih, w_hh, out, gates = ctx.saved_tensors
        B, T, I, ndir, H = ctx.dims
        dout = dout.contiguous().float()
        dgi = torch.empty((B, T, ndir, 3 * H), dtype=_f32, device=dout.device)
        dgh = torch.empty((B, T, ndir, 3 * H), dtype=_f32, device=dout.device)
        _lib.call("irx_gru_backward", _lib.ptr(dout), _lib.ptr(out), _lib.ptr(gates), _lib.ptr(lengths_i32),
                  _lib.ptr(w_hh), B, T, ndir, H, _lib.ptr(dgi), _lib.ptr(dgh), _lib.stream_ptr())
        dgi2 = dgi.view(B * T, ndir * 3 * H)
        dx = dgi2.mm(w_ih).view(B, T, I)
        dw_ih = dgi2.t().mm(x2)
        db_ih = dgi2.sum(0)
        # h_{t-1} in each direction's own order: forward = out shifted right, reverse = out shifted left
        o = out.view(B, T, ndir, H)
        hprev = torch.zeros_like(o)
        hprev[:, 1:, 0] = o[:, :-1, 0]
        if ndir == 2:
            hprev[:, :-1, 1] = o[:, 1:, 1]
        dw_hh = torch.einsum("btdg,btdh->dgh", dgh, hprev)
        db_hh = dgh.sum((0, 1))
        return dx, None, dw_ih, db_ih, dw_hh, db_hh


GRU_BACKEND = os.environ.get("IRX_GRU_BACKEND", "auto")     # "py": the Python autograd.Function below even when the C++ node exists


def gru_packed(gru: torch.nn.GRU, x, lengths, t_max):
    """x (B, >=t_max, I) on a HIP device; lengths (B,) int tensor on the same device. -> (B, t_max, ndir*H)."""
    assert gru.batch_first and gru.bias and gru.dropout == 0.0
    ndir = 2 if gru.bidirectional else 1
    len32 = lengths.to(torch.int32)
    h = x[:, :t_max].contiguous().float()
    nodes = None
    if h.is_cuda and GRU_BACKEND != "py":
        from . import _nodes
        nodes = _nodes.load()
    for layer in range(gru.num_layers):
        sfx = ["", "_reverse"][:ndir]
        if nodes is not None:
            # C++ autograd node (csrc/torch_nodes.cpp): nn.GRU's own parameter tensors go in (no cat / stack nodes), the weight
            # gradients come back through the optimizer's gradient sink when there is one
            params = [getattr(gru, "%s_l%d%s" % (n, layer, s_)) for s_ in sfx for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            slots, keep = (), ()
            if torch.is_grad_enabled():
                sink = getattr(params[0], "_irx_sink", None)
                if sink is not None:
                    ent = sink[0].native_sink(("gru", id(params[0])), params)
                    if ent is not None:
                        slots, keep = ent
            h = nodes.gru_layer(h, len32, params, _lib.stream_ptr(), slots, keep)
            continue
        w_ih = torch.cat([getattr(gru, "weight_ih_l%d%s" % (layer, s)) for s in sfx], 0)
        b_ih = torch.cat([getattr(gru, "bias_ih_l%d%s" % (layer, s)) for s in sfx], 0)
        w_hh = torch.stack([getattr(gru, "weight_hh_l%d%s" % (layer, s)) for s in sfx], 0)
        b_hh = torch.stack([getattr(gru, "bias_hh_l%d%s" % (layer, s)) for s in sfx], 0)
        h = GRULayerFn.apply(h, len32, w_ih, b_ih, w_hh, b_hh)
    return h


class CosineRowsFn(torch.autograd.Function):
    """score[i] = cos(a_i, b_{idx[i]}) with per-vector norm clamps (F.normalize / F.cosine_similarity semantics):
    one launch forward, one backward pair (csrc/irx_match.hip) instead of ~27 ATen ops per matching head."""

    @staticmethod
    def forward(ctx, a, b, idx, eps):
        a = a.contiguous().float()
        b = b.contiguous().float()
        n, d = a.shape
        score = torch.empty(n, dtype=_f32, device=a.device)
        norms = torch.empty((max(n, 1), 2), dtype=_f32, device=a.device)
        _lib.call("irx_cosine_rows_fwd", _lib.ptr(a), _lib.ptr(b), _lib.ptr(idx) if idx is not None else None, n, d,
                  float(eps), _lib.ptr(score), _lib.ptr(norms), _lib.stream_ptr())
        ctx.save_for_backward(a, b, idx, score, norms)
        ctx.eps = float(eps)
        return score

    @staticmethod
    def backward(ctx, dscore):
        a, b, idx, score, norms = ctx.saved_tensors
        n, d = a.shape
        m = b.shape[0]
        dscore = dscore.contiguous().float()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.call("irx_cosine_rows_bwd", _lib.ptr(a), _lib.ptr(b), _lib.ptr(idx) if idx is not None else None,
                  _lib.ptr(score), _lib.ptr(norms), _lib.ptr(dscore), n, m, d, ctx.eps, _lib.ptr(da), _lib.ptr(db),
                  _lib.stream_ptr())
        return da, db, None, None


def cosine_rows(a, b, idx=None, eps=1e-8):
    """Row-wise cosine of a (n, d) against b[idx] ((m, d), idx int64 (n,) or None). HIP tensors -> the fused kernels;
    host tensors (CPU unit tests of the head logic) -> the PyTorch formulation."""
    if a.is_cuda:
        return CosineRowsFn.apply(a, b, idx, eps)
    bb = b if idx is None else b.index_select(0, idx)
    return torch.nn.functional.cosine_similarity(a, bb, dim=1, eps=eps)


class ContrastiveFn(torch.autograd.Function):
    """sum over scored scenes of keep_s * clamp(logsumexp(x * (1 - lab)) - sum(x * lab) + margin, 0), x = gamma*(s1+s2+s3):
    the reference's per-sample ContrastiveLoss (lib/loss_helper.py:93-107) for all scenes in one launch each way."""

    @staticmethod
    def forward(ctx, s1, s2, s3, lab, seg_off, keep, gamma, margin):
        s1, s2, s3 = s1.contiguous().float(), s2.contiguous().float(), s3.contiguous().float()
        nseg = keep.shape[0]
        dev = s1.device
        out = torch.empty(1, dtype=_f32, device=dev)
        scratch = torch.empty((3, max(nseg, 1)), dtype=_f32, device=dev)
        _lib.call("irx_contrastive_fwd", _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(s3), _lib.ptr(lab), _lib.ptr(seg_off),
                  _lib.ptr(keep), nseg, float(gamma), float(margin), _lib.ptr(out), scratch[0].data_ptr(),
                  scratch[1].data_ptr(), scratch[2].data_ptr(), _lib.stream_ptr())
        ctx.save_for_backward(s1, s2, s3, lab, seg_off, scratch)
        ctx.gamma = float(gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        s1, s2, s3, lab, seg_off, scratch = ctx.saved_tensors
        nseg = seg_off.shape[0] - 1
        ds = torch.zeros_like(s1)
        dout = dout.contiguous().float()
        _lib.call("irx_contrastive_bwd", _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(s3), _lib.ptr(lab), _lib.ptr(seg_off),
                  scratch[1].data_ptr(), scratch[2].data_ptr(), _lib.ptr(dout), nseg, ctx.gamma, _lib.ptr(ds),
                  _lib.stream_ptr())
        return ds, ds, ds, None, None, None, None, None



class LangPoolFn(torch.autograd.Function):
    """The language module's four attention heads (reference models/lang_module.py:61-83) in one launch each way
    (irx_lang_pool_fwd / _bwd, csrc/irx_match.hip): (feats (B, T, O), embed (B, T, E), lengths (B,), w0, b0, ..., w3, b3) ->
    (att (B, T, 4), pooled (B, 4, E)) with softmax over ALL T positions, then mask + renormalise. The heads' own nn.Linear
    parameters go in one by one (no cat node between them and this one)."""

    @staticmethod
    def forward(ctx, feats, embed, length, *wb):
        import ctypes
        feats, embed = feats.contiguous().float(), embed.contiguous().float()
        length = length.contiguous().to(torch.int64)
        B, T, O = feats.shape
        E = embed.shape[2]
        dev = feats.device
        ws = [w.contiguous() for w in wb[0::2]]
        bs = [b.contiguous() for b in wb[1::2]]
        att = torch.empty((B, T, 4), dtype=_f32, device=dev)
        prob = torch.empty((B, T, 4), dtype=_f32, device=dev)
        qsum = torch.empty((B, 4), dtype=_f32, device=dev)
        pooled = torch.empty((B, 4, E), dtype=_f32, device=dev)
        wp = (ctypes.c_void_p * 4)(*[w.data_ptr() for w in ws])
        bp = (ctypes.c_void_p * 4)(*[b.data_ptr() for b in bs])
        _lib.call("irx_lang_pool_fwd", _lib.ptr(feats), _lib.ptr(embed), _lib.ptr(length), B, T, O, E, wp, bp, _lib.ptr(att),
                  _lib.ptr(prob), _lib.ptr(qsum), _lib.ptr(pooled), _lib.stream_ptr())
        ctx.save_for_backward(feats, embed, length, att, prob, qsum, *ws)
        return att, pooled

    @staticmethod
    def backward(ctx, datt, dpooled):
        import ctypes
        feats, embed, length, att, prob, qsum, *ws = ctx.saved_tensors
        B, T, O = feats.shape
        E = embed.shape[2]
        dev = feats.device
        dpooled = dpooled.contiguous().float() if dpooled is not None else torch.zeros((B, 4, E), dtype=_f32, device=dev)
        datt = datt.contiguous().float() if datt is not None else None
        dfeats, dembed = torch.empty_like(feats), torch.empty_like(embed)
        dwb = torch.empty(4 * O + 4, dtype=_f32, device=dev)
        part = torch.empty(B * 4 * (O + 1), dtype=_f32, device=dev)
        wp = (ctypes.c_void_p * 4)(*[w.data_ptr() for w in ws])
        _lib.call("irx_lang_pool_bwd", _lib.ptr(feats), _lib.ptr(embed), _lib.ptr(length), B, T, O, E, wp, _lib.ptr(att), _lib.ptr(prob),
                  _lib.ptr(qsum), _lib.ptr(dpooled), _lib.ptr(datt), _lib.ptr(dfeats), _lib.ptr(dembed), dwb.data_ptr(),
                  dwb.data_ptr() + 16 * O, _lib.ptr(part), _lib.stream_ptr())
        grads = []
        for h in range(4):
            grads += [dwb[h * O:(h + 1) * O].view(1, O), dwb[4 * O + h:4 * O + h + 1]]
        return (dfeats, dembed, None) + tuple(grads)


class AttentionPoolFn(torch.autograd.Function):
    """(feats (B, n, D), lang (B, D)) -> (atten (B, n) = softmax_i(<feats_i, lang> / sqrt(D)), pooled (B, D) = sum_i atten_i feats_i):
    the scene head's language-guided attention (reference models/scene_module.py:84-93) in one launch each way
    (irx_attn_pool_fwd / _bwd, csrc/irx_match.hip) instead of bmm, div, softmax, mul, sum and their ~10 backward launches."""

    @staticmethod
    def forward(ctx, feats, lang):
        feats, lang = feats.contiguous().float(), lang.contiguous().float()
        B, n, d = feats.shape
        atten = torch.empty((B, n), dtype=_f32, device=feats.device)
        out = torch.empty((B, d), dtype=_f32, device=feats.device)
        ctx.scale = 1.0 / math.sqrt(d)
        _lib.call("irx_attn_pool_fwd", _lib.ptr(feats), _lib.ptr(lang), B, n, d, ctx.scale, _lib.ptr(atten), _lib.ptr(out),
                  _lib.stream_ptr())
        ctx.save_for_backward(feats, lang, atten)
        return atten, out

    @staticmethod
    def backward(ctx, datten, dout):
        feats, lang, atten = ctx.saved_tensors
        B, n, d = feats.shape
        dfeats = torch.empty_like(feats)
        dlang = torch.empty_like(lang)
        dout = dout.contiguous().float() if dout is not None else torch.zeros((B, d), dtype=_f32, device=feats.device)
        datten = datten.contiguous().float() if datten is not None else None
        _lib.call("irx_attn_pool_bwd", _lib.ptr(feats), _lib.ptr(lang), _lib.ptr(atten), _lib.ptr(dout), _lib.ptr(datten), B, n, d,
                  ctx.scale, _lib.ptr(dfeats), _lib.ptr(dlang), _lib.stream_ptr())
        return dfeats, dlang


class TotalLossFn(torch.autograd.Function):
    """get_loss's arithmetic in ONE launch, gradients included (irx_total_loss, csrc/irx_match.hip): language cross-entropy,
    area cross-entropy + accuracy, batched ContrastiveLoss / batch_size and their weighted sum (reference lib/loss_helper.py:
    93-118,121-150,248-263). -> (loss (1,), ref_loss (1,), lang_loss (), seg_loss (), seg_acc ()); only `loss` is differentiable.
    The backward is one multiply of the stored gradients by the upstream gradient."""

    @staticmethod
    def forward(ctx, lang_scores, seg_scores, s1, s2, s3, lang_label, seg_label, lab, seg_off, keep, gamma, margin, ref_weight,
                batch_size):
        lang_scores, seg_scores = lang_scores.contiguous().float(), seg_scores.contiguous().float()
        s1, s2, s3 = s1.contiguous().float(), s2.contiguous().float(), s3.contiguous().float()
        B, n_lang = lang_scores.shape
        n_seg = seg_scores.shape[1]
        nscored, ns = int(keep.shape[0]), int(s1.shape[0])
        dev = lang_scores.device
        out = torch.empty(5, dtype=_f32, device=dev)
        grads = torch.zeros(B * n_lang + B * n_seg + ns, dtype=_f32, device=dev) if nscored == 0 else \
            torch.empty(B * n_lang + B * n_seg + ns, dtype=_f32, device=dev)
        o1, o2 = B * n_lang, B * n_lang + B * n_seg
        _lib.call("irx_total_loss", _lib.ptr(lang_scores), _lib.ptr(lang_label), B, n_lang, _lib.ptr(seg_scores), _lib.ptr(seg_label),
                  n_seg, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(s3), _lib.ptr(lab), _lib.ptr(seg_off), _lib.ptr(keep), nscored,
                  float(gamma), float(margin), float(ref_weight), int(batch_size), _lib.ptr(out), grads.data_ptr(),
                  grads.data_ptr() + 4 * o1, grads.data_ptr() + 4 * o2, _lib.stream_ptr())
        ctx.save_for_backward(grads)
        ctx.dims = (B, n_lang, n_seg, ns)
        loss, ref, lang, seg, acc = out[0:1], out[1:2], out[2], out[3], out[4]
        ctx.mark_non_differentiable(ref, lang, seg, acc)
        return loss, ref, lang, seg, acc

    @staticmethod
    def backward(ctx, g, *_):
        (grads,) = ctx.saved_tensors
        B, n_lang, n_seg, ns = ctx.dims
        scaled = grads * g.reshape(1).to(grads.dtype)
        o1, o2 = B * n_lang, B * n_lang + B * n_seg
        ds = scaled[o2:o2 + ns]
        return (scaled[:o1].view(B, n_lang), scaled[o1:o2].view(B, n_seg), ds, ds, ds) + (None,) * 9


# ---------------------------------------------------------------------------------------------------------------------
# The head MLPs nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear) as ONE autograd node (csrc/irx_mlp.hip)
class MLP2Fn(torch.autograd.Function):
    """y = W2 . D(relu(N(W1 x + b1))) + b2 — irx_mlp2_fwd / irx_mlp2_bwd (include/irx.h): one C-ABI call each way instead of
    ~8 forward and ~25 backward ATen dispatches. norm: 1 BatchNorm1d (batch statistics), 2 BatchNorm1d (running statistics),
    3 LayerNorm."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2, norm, eps, rmean, rvar, momentum, drop_p, seed):
        x = x.contiguous().float()
        rows, din = x.shape
        dh, dout = w1.shape[0], w2.shape[0]
        y = torch.empty((rows, dout), dtype=_f32, device=x.device)
        saved = torch.empty(int(_lib.load().irx_mlp2_saved_floats(rows, dh)), dtype=_f32, device=x.device)
        _lib.call("irx_mlp2_fwd", _lib.ptr(x), rows, din, dh, dout, _lib.ptr(w1), _lib.ptr(b1), norm, _lib.ptr(gamma),
                  _lib.ptr(beta), float(eps), _lib.ptr(rmean), _lib.ptr(rvar), float(momentum), float(drop_p), int(seed),
                  _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(saved), _lib.ptr(y), _lib.stream_ptr())
        ctx.save_for_backward(x, w1, gamma, w2, saved)
        ctx.cfg = (norm, float(drop_p))
        params = (w1, b1, gamma, beta, w2, b2)
        ctx.params = params if all(getattr(p, "_irx_sink", None) is not None for p in params) else None
        ctx.key = ("mlp2", id(w1))
        ctx.stream = torch.cuda.current_stream()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, gamma, w2, saved = ctx.saved_tensors
        norm, drop_p = ctx.cfg
        rows, din = x.shape
        dh, dout = w1.shape[0], w2.shape[0]
        dy = dy.contiguous().float()
        want_dx = ctx.needs_input_grad[0]
        # Gradient sink (optim.FlatAdam): the six parameter gradients go straight into the optimizer's flat gradient buffer —
        # no AccumulateGrad nodes, no .grad tensors, no copy at gather time. Only when this node runs on the optimizer's
        # stream (the heads do: autograd replays them on the main stream) and every parameter is the optimizer's.
        slots = None
        sink = getattr(w1, "_irx_sink", None)
        if sink is not None and ctx.params is not None and torch.cuda.current_stream() == ctx.stream:
            slots = sink[0].sink_slots(ctx.key, ctx.params)
        scratch = torch.empty(rows * (dh + (din if want_dx else 0)), dtype=_f32, device=x.device)
        base = scratch.data_ptr()
        dx_ptr = base + 4 * rows * dh if want_dx else None
        if slots is not None:
            gp = [t.data_ptr() for t in slots]           # order of ctx.params: w1, b1, gamma, beta, w2, b2
            grads = (None,) * 6
        else:
            sizes = (dh * din, dh, dh, dh, dout * dh, dout)
            buf = torch.empty(sum(sizes), dtype=_f32, device=x.device)
            gp, grads, off = [], [], 0
            for n, shape in zip(sizes, ((dh, din), (dh,), (dh,), (dh,), (dout, dh), (dout,))):
                gp.append(buf.data_ptr() + 4 * off)
                grads.append(buf[off:off + n].view(shape))
                off += n
        _lib.call("irx_mlp2_bwd", x.data_ptr(), dy.data_ptr(), rows, din, dh, dout, w1.data_ptr(), norm, gamma.data_ptr(),
                  w2.data_ptr(), saved.data_ptr(), 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0, base, dx_ptr,
                  gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], _lib.stream_ptr())
        if slots is not None:
            sink[0].sink_delivered_inline(ctx.key, ctx.params)
        dx = scratch[rows * dh:].view(rows, din) if want_dx else None
        return (dx,) + tuple(grads) + (None,) * 7


# Two ways to run the operator. As the Python node above it is a measured NEGATIVE result (round 4, pinned host, bf16 B = 16,
# two alternating pairs: 6.49 / 6.50 ms per step with the seven head MLPs on it against 6.15 / 6.13 ms through ATen): the step
# is host-bound there, and a Python autograd.Function costs more interpreter time (31 us forward + 97 us backward per MLP in
# torch.profiler) than the ~8 + ~25 ATen dispatches it replaces, which run inside the C++ engine without the interpreter.
# As a C++ autograd node (csrc/torch_nodes.cpp, loaded by _nodes.py) the same two C-ABI calls cost a few microseconds each
# way, so that is the default whenever the extension has been built; IRX_FUSED_MLP2=0 keeps ATen, =1 forces the operator
# (through the Python node if the extension is missing), IRX_MLP2_BACKEND=py selects the Python node for tests.
FUSED_MLP2 = {"0": False, "1": True}.get(os.environ.get("IRX_FUSED_MLP2", "auto"))     # None: on iff the C++ node is there
MLP2_BACKEND = os.environ.get("IRX_MLP2_BACKEND", "auto")


def _mlp2_backend():
    """-> the C++ extension module, "py" (MLP2Fn) or None (ATen modules)"""
    if FUSED_MLP2 is False:
        return None
    if MLP2_BACKEND != "py":
        from . import _nodes
        mod = _nodes.load()
        if mod is not None:
            return mod
    return "py" if FUSED_MLP2 else None


def mlp_relu2(seq, x):
    """Apply `seq` = nn.Sequential(Linear, ReLU, Dropout, Linear, ReLU) (the language module's word projection, reference
    models/lang_module.py:33-37) to (..., C) device rows through the fused operator's no-normalisation variant (irx_mlp2_fwd with
    norm = 4 | 8) as ONE C++ autograd node — 2 launches instead of 5 ATen operators forward, 3 instead of ~12 backward, parameter
    gradients through the optimizer's sink. Anything else (host tensors, no C++ nodes module, another layout) goes through the
    module itself. Same parameters and state-dict keys either way.
    NOT used by LangModule: a measured negative result at the model's size (480 rows: 5.98 vs 5.64 ms per step) — the
    operator's 64-row tiles suit the 16..64-row heads, not a 480 x 300 x 256 GEMM; kept as a tested operator."""
    import torch.nn as nn
    mods = list(seq)
    backend = _mlp2_backend()
    ok = (backend is not None and backend != "py" and x.is_cuda and x.shape[-1] > 0 and x.numel() > 0 and len(mods) == 5
          and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.ReLU) and isinstance(mods[2], nn.Dropout)
          and isinstance(mods[3], nn.Linear) and isinstance(mods[4], nn.ReLU) and mods[0].bias is not None and mods[3].bias is not None)
    if not ok:
        return seq(x)
    lin1, lin2 = mods[0], mods[3]
    drop_p = mods[2].p if mods[2].training else 0.0
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    seed = _dropout_seed(x.device) if drop_p > 0 else 0
    slots, keep = (), ()
    if torch.is_grad_enabled():
        sink = getattr(lin1.weight, "_irx_sink", None)
        if sink is not None:
            ent = sink[0].native_sink(("mlp_relu2", id(lin1.weight)), (lin1.weight, lin1.bias, lin2.weight, lin2.bias))
            if ent is not None:
                slots, keep = ent
    y = backend.mlp_relu2(x2, lin1.weight, lin1.bias, lin2.weight, lin2.bias, drop_p, seed if seed < (1 << 63) else seed - (1 << 64),
                          _lib.stream_ptr(), slots, keep)
    return y.view(*lead, lin2.out_features)


_SEED_LOCK = __import__('threading').Lock()


def _dropout_seed(device):
    """A 64-bit key for one dropout call, drawn host-side from the device's default generator: (seed, Philox offset), and the
    offset is advanced like a real dropout kernel would — so torch.manual_seed() reproduces the masks, two calls never share
    one, and no GPU work or sync is involved."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    with _SEED_LOCK:                                   # (the helper thread draws too: read-modify-write of the offset)
        off = gen.get_offset()
        gen.set_offset(off + 4)
    return (gen.initial_seed() * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03 + 1) & 0xFFFFFFFFFFFFFFFF


def mlp2(seq, x, seed=None):
    """Apply a head MLP `seq` = nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear) to (rows, C) device rows
    through the fused operator; anything else (host tensors, another layout, sync-BatchNorm over more than one rank, a
    single row in training) goes through the module itself. Same parameters, buffers and state-dict keys either way."""
    import torch.nn as nn
    mods = list(seq)
    backend = _mlp2_backend()
    ok = (backend is not None and x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and len(mods) in (4, 5) and isinstance(mods[0], nn.Linear)
          and isinstance(mods[1], (nn.BatchNorm1d, nn.LayerNorm)) and isinstance(mods[2], nn.ReLU) and isinstance(mods[-1], nn.Linear)
          and (len(mods) == 4 or isinstance(mods[3], nn.Dropout)) and mods[0].bias is not None and mods[-1].bias is not None)
    if ok and isinstance(mods[1], nn.BatchNorm1d):
        bn = mods[1]
        if getattr(bn, "_irx_sync", False):
            import torch.distributed as dist
            ok = not (bn.training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        ok = ok and bn.affine and bn.track_running_stats and bn.momentum is not None and not (bn.training and x.shape[0] < 2)
    elif ok:
        ln = mods[1]
        ok = ln.elementwise_affine and len(ln.normalized_shape) == 1
    if not ok:
        return seq(x)
    lin1, nrm, lin2 = mods[0], mods[1], mods[-1]
    drop_p = mods[3].p if (len(mods) == 5 and mods[3].training) else 0.0
    if isinstance(nrm, nn.BatchNorm1d):
        norm = 1 if nrm.training else 2
        rmean, rvar, momentum = nrm.running_mean, nrm.running_var, nrm.momentum
        if nrm.training:
            from . import _counters
            _counters.bump([nrm.num_batches_tracked])
    else:
        norm, rmean, rvar, momentum = 3, None, None, 0.0
    if seed is None or drop_p <= 0:                      # (seed: a caller that draws its seeds up front, heads.draw_seeds)
        seed = _dropout_seed(x.device) if drop_p > 0 else 0
    if backend != "py":
        # C++ node: the optimizer's slot addresses (gradient sink) are looked up once per MLP and travel as integers
        slots, keep = (), ()
        if torch.is_grad_enabled():
            sink = getattr(lin1.weight, "_irx_sink", None)
            if sink is not None:
                ent = sink[0].native_sink(("mlp2", id(lin1.weight)),
                                          (lin1.weight, lin1.bias, nrm.weight, nrm.bias, lin2.weight, lin2.bias))
                if ent is not None:
                    slots, keep = ent
        return backend.mlp2(x, lin1.weight, lin1.bias, nrm.weight, nrm.bias, lin2.weight, lin2.bias, norm, nrm.eps, rmean, rvar,
                            momentum, drop_p, seed if seed < (1 << 63) else seed - (1 << 64), _lib.stream_ptr(), slots, keep)
    return MLP2Fn.apply(x, lin1.weight, lin1.bias, nrm.weight, nrm.bias, lin2.weight, lin2.bias, norm, nrm.eps, rmean, rvar,
                        momentum, drop_p, seed)
