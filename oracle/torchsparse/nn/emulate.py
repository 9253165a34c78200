"""Emulation of the BUILD's bf16 arithmetic inside the CPU oracle — TEST INFRASTRUCTURE (see oracle/__init__.py).

This is NOT upstream torchsparse behaviour: the reference computes everything in fp32 (models/basic_blocks.py:59-95).
BASELINE configs[2]-[4] name bf16, and libirx.so offers two modes for it (include/irx.h, irx_set_compute_dtype):

  "bf16_operands" (1)  every 32/64/128-channel sparse-conv launch (k_spconv2: forward + data-gradient, k_wgrad_pairs:
                       weight-gradient through pair lists) rounds x, w and dy to bf16 (round-to-nearest-even) as they enter
                       the matrix core and accumulates in fp32; every tensor in HBM stays fp32. The small-Cin stem kernels
                       are fp32; the multiview stem (Cin 129..136) rounds its 128 leading channels only. Weight-gradients
                       WITHOUT pair lists (the scene head's BEV / Conv2d-as-sparse-conv, k_spconv2_wgrad) stay fp32.
  "bf16" (2)           as (1), and INSIDE the one-call encoder executor (training mode) every conv output c_i, layer output
                       y_i and gradient in flight (d y_i, d c_i, shortcut gradient) is a bf16 array in HBM — i.e. it is
                       rounded once when it is stored. The encoder's input features, its final output and that output's
                       gradient, BatchNorm statistics, parameters and parameter gradients stay fp32. The multiview stem
                       stores its tail-channel partial sum (bf16) and then adds the main channels on top (second rounding).

`mode(...)` selects what the oracle's Conv3d / ReLU / `+` and model_ref's scene head do; `encoder_scope(n_relu)` marks
one executor pass (the last of its n_relu ReLU sites is the fp32 output). With mode None (default) nothing changes.
Only the rounding POINTS are restated here; the fp32 summation order inside a kernel is not. What that allows (measured,
tools/bf16_emul_diag.py): ONE layer fed with bit-identical stored inputs agrees to ~3e-5 relative L2 — a value within fp32
round-off (1e-7) of a bf16 tie is rounded the other way with probability 1e-7 / 6e-3, a full bf16 ulp each time, i.e.
sqrt(6e-3 * 1e-7). But rounding is DISCONTINUOUS: an input deviation e becomes sqrt(6e-3 * e) after the next rounding, so
through a chain of layers 1e-7 -> 3e-5 -> 4e-4 -> 1.5e-3 -> ... converges to the bf16 noise floor (6e-3) within ~8 layers
for ANY two implementations that do not sum in exactly the same order — including this emulation run twice with fp32 and
with float64 accumulation (`acc64`). Hence the two kinds of test in tests/test_bf16_gpu.py: (a) layer by layer with the
executor's own stored tensors as inputs ("teacher forced"), where the bar is 2e-4; (b) end to end, where the bar is the
emulation's own reordering distance (HIP vs emulation <= 2 x emulation(fp32 sums) vs emulation(float64 sums))."""
import contextlib

import torch

MODE = None            # None | "bf16_operands" | "bf16"
WIDE_STEM_PAIR_LISTS = True   # the executor's multiview stem takes the pair-list (bf16) weight-gradient; the per-layer op does not
ACC64 = False          # accumulate the conv sums in float64 (then round to fp32): a second, equally valid summation order
_SCOPE = []            # stack of {"n": relu sites of the encoder, "seen": ...}

FAST = (32, 64, 128)


def rb(t):
    """fp32 -> bf16 (round to nearest even, what v_cvt_pk_bf16_f32 does) -> fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


@contextlib.contextmanager
def mode(name):
    global MODE
    assert name in (None, "fp32", "bf16_operands", "bf16")
    saved, MODE = MODE, (None if name == "fp32" else name)
    try:
        yield
    finally:
        MODE = saved


@contextlib.contextmanager
def per_layer_ops():
    """The per-layer C-ABI path (sparse/functional.SparseConvFn, no executor): the multiview stem's weight-gradient runs on
    the offset-major fp32 kernel there (no pair lists are handed to irx_spconv_wgrad)."""
    global WIDE_STEM_PAIR_LISTS
    saved, WIDE_STEM_PAIR_LISTS = WIDE_STEM_PAIR_LISTS, False
    try:
        yield
    finally:
        WIDE_STEM_PAIR_LISTS = saved


@contextlib.contextmanager
def acc64(on=True):
    global ACC64
    saved, ACC64 = ACC64, bool(on)
    try:
        yield
    finally:
        ACC64 = saved


@contextlib.contextmanager
def encoder_scope(n_relu, active=True):
    """One pass of the encoder executor: storage rounding applies between its input and its last ReLU."""
    if MODE != "bf16" or not active:
        yield
        return
    _SCOPE.append({"n": int(n_relu), "seen": 0})
    try:
        yield
    finally:
        _SCOPE.pop()


def storing():
    return MODE == "bf16" and bool(_SCOPE)


class _Q(torch.autograd.Function):
    """identity with optional bf16 rounding of the value (forward) and of the TOTAL gradient of the value (backward)"""

    @staticmethod
    def forward(ctx, t, fwd, bwd):
        ctx.bwd = bwd
        return rb(t) if fwd else t.clone()

    @staticmethod
    def backward(ctx, g):
        return (rb(g) if ctx.bwd else g), None, None


def q(t, fwd, bwd):
    return _Q.apply(t, bool(fwd), bool(bwd))


def on_relu(out):
    """y_i = ReLU(...) of a layer: stored as bf16 (and its gradient d y_i too) unless it is the executor's output."""
    if not storing():
        return out
    s = _SCOPE[-1]
    s["seen"] += 1
    if s["seen"] >= s["n"]:
        return out
    return q(out, True, True)


def on_shortcut(feats):
    """the shortcut operand of `net(x) + x`: its gradient (dres) is stored as bf16 before the main-path gradient of the
    next conv is accumulated on top (which is rounded again by the q() of y_i)."""
    return q(feats, False, True) if storing() else feats


def _acc(t):
    return t.double() if ACC64 else t


def _gemm_fwd(x, w, maps, n_out):
    x, w = _acc(x), _acc(w)
    out = torch.zeros(n_out, w.shape[-1], dtype=x.dtype)
    for k, (i_idx, o_idx) in enumerate(maps):
        if i_idx.numel():
            out.index_add_(0, o_idx, x.index_select(0, i_idx).mm(w[k]))
    return out.float()


class _Conv(torch.autograd.Function):
    """out[o] += x[i] @ w[k] over the pairs of every offset, with per-use operand rounding:
    flags = (fwd, dgrad, wgrad): round both operands of that pass to bf16 (fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x, w, maps, n_out, flags):
        ctx.maps, ctx.flags, ctx.n_in = maps, flags, x.shape[0]
        ctx.save_for_backward(x, w)
        if flags[0]:
            x, w = rb(x), rb(w)
        return _gemm_fwd(x, w, maps, n_out)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        _, dgrad, wgrad = ctx.flags
        dx = dw = None
        if ctx.needs_input_grad[0]:
            gq, wq = (rb(g), rb(w)) if dgrad else (g, w)
            gq, wq = _acc(gq), _acc(wq)
            dx = torch.zeros(ctx.n_in, w.shape[1], dtype=gq.dtype)
            for k, (i_idx, o_idx) in enumerate(ctx.maps):
                if i_idx.numel():
                    dx.index_add_(0, i_idx, gq.index_select(0, o_idx).mm(wq[k].t()))
            dx = dx.float()
        if ctx.needs_input_grad[1]:
            gq, xq = (rb(g), rb(x)) if wgrad else (g, x)
            gq, xq = _acc(gq), _acc(xq)
            dw = torch.zeros(w.shape, dtype=gq.dtype)
            for k, (i_idx, o_idx) in enumerate(ctx.maps):
                if i_idx.numel():
                    dw[k] = xq.index_select(0, i_idx).t().mm(gq.index_select(0, o_idx))
            dw = dw.float()
        return dx, dw, None, None, None


def conv(features, kernel, maps, n_out, pair_lists=True):
    """The sparse conv of one layer under the current mode. pair_lists: the layer's weight-gradient runs through pair lists
    (every encoder layer); False = the scene head's tables (fp32 weight-gradient kernel)."""
    K, cin, cout = kernel.shape
    st = storing()
    if cin in FAST and cout in FAST:
        out = _Conv.apply(features, kernel, maps, n_out, (True, True, bool(pair_lists)))
    elif cin <= 8 and cout == 32:
        out = _Conv.apply(features, kernel, maps, n_out, (False, False, False))          # stem: fp32 kernels
    elif K == 27 and 128 < cin <= 136 and cout == 32:
        # multiview stem: tail channels on the fp32 stem kernels FIRST (stored), main 128 channels added on top
        tail = _Conv.apply(features[:, 128:], kernel[:, 128:], maps, n_out, (False, False, False))
        main = _Conv.apply(features[:, :128], kernel[:, :128], maps, n_out, (True, True, bool(pair_lists and WIDE_STEM_PAIR_LISTS)))
        if st:
            tail = q(tail, True, False)
        out = tail + main
    else:
        raise NotImplementedError("no bf16 path for a %d -> %d channel conv (the build has none either)" % (cin, cout))
    # conv output c_i and its gradient d c_i are bf16 arrays inside the executor
    return q(out, True, True) if st else out


class _Conv2d(torch.autograd.Function):
    """nn.Conv2d of the scene head = k_spconv2 over a constant 9-offset table: forward and data-gradient with bf16
    operands, weight-gradient by the fp32 offset-major kernel (no pair lists for these tables)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.conv2d(rb(x), rb(w))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = torch.nn.grad.conv2d_input(x.shape, rb(w), rb(g)) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(x, w.shape, g) if ctx.needs_input_grad[1] else None
        return dx, dw


def conv2d(module, x):
    if MODE is None or module.in_channels not in FAST or module.out_channels not in FAST:
        return module(x)
    y = _Conv2d.apply(x, module.weight)
    return y if module.bias is None else y + module.bias.view(1, -1, 1, 1)
