"""Raw operator wrappers over libirx.so plus the autograd Functions built from them.

Each Function stands in for one torchsparse / torch_scatter operator the reference reaches from
models/basic_blocks.py (reference tree); see include/irx.h for the per-symbol call sites.
Everything here requires a HIP device: there is no CPU fallback (see _lib.py).
"""
import torch

from .. import _lib

# bench.py sets PROFILE to a list to collect (kind, n_out, K, cin, cout, M, start_event, end_event) per sparse-conv
# launch, with events recorded on the launch stream. None (default) = no instrumentation at all.
PROFILE = None
PROFILE_NO_COUNT = False     # bench.py's timeline mode: no pair counting (it syncs)
_PAIR_COUNT = {}

_i32 = torch.int32
_i64 = torch.int64
_f32 = torch.float32


RANGE_ERROR = ("voxel coordinates outside [-32768, 32768) or more than 32767 batch items: such points cannot be keyed "
               "(irx_quantize marks them and the voxel count comes back negative); check the voxel size / units")


def _stream():
    return _lib.stream_ptr()


def _bracket():
    """Two timing events that the library records right around the next dominant sparse-conv kernel of this thread
    (irx_profile_next_kernel): the measured span excludes the weight-permute / split-reduce helper launches."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()                                      # instantiates the HIP events; re-recorded by the library
    e1.record()
    _lib.call("irx_profile_next_kernel", e0.cuda_event, e1.cuda_event)
    return e0, e1


def _f32c(t):
    if t.dtype != _f32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------- coordinates --
def coords_to_keys(coords):
    n = coords.shape[0]
    keys = torch.empty(n, dtype=_i64, device=coords.device)
    _lib.call("irx_coords_to_keys", _lib.ptr(coords), n, _lib.ptr(keys), _stream())
    return keys


def keys_to_coords(keys):
    """(n,) int64 Morton keys -> (n, 4) int32 coordinate rows (x, y, z, batch): irx_keys_to_coords."""
    n = keys.shape[0]
    coords = torch.empty((n, 4), dtype=_i32, device=keys.device)
    _lib.call("irx_keys_to_coords", _lib.ptr(keys), n, _lib.ptr(coords), _stream())
    return coords


def sort_keys(keys, end_bit=63, n_dev=None, pad=None, begin_bit=0):
    """Stable ascending sort of int64 `keys` (non-negative) by bits [begin_bit, end_bit) -> (sorted keys, order int32) with
    sorted[i] = keys[order[i]]: irx_sort_pairs_u64 (csrc/irx_sort.hip), the build's own radix sort. n_dev (int32 device
    tensor, 1 element) + pad: only the first n_dev keys are real, the rest sorts behind them as `pad` (no host sync)."""
    n = keys.shape[0]
    dev = keys.device
    out = torch.empty(n, dtype=_i64, device=dev)
    order = torch.empty(n, dtype=_i32, device=dev)
    if n == 0:
        return out, order
    keys = keys.contiguous()
    wsb = int(_lib.load().irx_sort_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.call("irx_sort_pairs_u64", _lib.ptr(keys), n, _lib.ptr(n_dev), int(pad or 0), int(begin_bit), int(end_bit), _lib.ptr(out),
              _lib.ptr(order), _lib.ptr(ws), wsb, _stream())
    return out, order


def morton_bits(batch_size, with_pad=False):
    """Bits of a Morton key (batch << 48 | 48-bit interleave) that can differ for batch indices < batch_size (<= batch_size
    when a padding key `batch_size << 48` is in play); None = unknown batch size -> all 63."""
    if batch_size is None:
        return 63
    top = int(batch_size) if with_pad else max(int(batch_size) - 1, 0)
    return 48 + max(top.bit_length(), 1)


def quantize(xyz, batch, voxel, want_coords=True):
    """xyz (N,3) f32/f64 cuda, batch (N,) int32 or None, voxel: 3 floats -> coords (N,4) i32 (None when not wanted: the
    voxeliser decodes the rows it keeps from their keys), keys (N,) i64."""
    n = xyz.shape[0]
    xyz = xyz.contiguous()
    assert xyz.dtype in (torch.float32, torch.float64)
    coords = torch.empty((n, 4), dtype=_i32, device=xyz.device) if want_coords else None
    keys = torch.empty(n, dtype=_i64, device=xyz.device)
    _lib.call("irx_quantize", _lib.ptr(xyz), int(xyz.dtype == torch.float64), _lib.ptr(batch), n,
              float(voxel[0]), float(voxel[1]), float(voxel[2]), _lib.ptr(coords), _lib.ptr(keys), _stream())
    return coords, keys


def new_table(n, device):
    cap = _lib.hash_capacity(n)
    return (torch.empty(cap, dtype=_i64, device=device), torch.empty(cap, dtype=_i32, device=device), cap)


def voxel_unique(keys):
    """First-occurrence voxel de-duplication. Returns int64 indices (unordered) of the winning points."""
    n = keys.shape[0]
    tk, tv, cap = new_table(n, keys.device)
    _lib.call("irx_voxel_insert", _lib.ptr(keys), n, _lib.ptr(tk), _lib.ptr(tv), cap, _stream())
    winners = torch.empty(max(n, 1), dtype=_i32, device=keys.device)
    count = torch.empty(1, dtype=_i32, device=keys.device)
    _lib.call("irx_voxel_select", _lib.ptr(keys), n, _lib.ptr(tk), _lib.ptr(tv), cap, _lib.ptr(winners),
              _lib.ptr(count), _stream())
    m = int(count.item())  # host sync: the voxel count sizes every later buffer
    if m < 0:
        raise ValueError(RANGE_ERROR)
    return winners[:m].long()


def voxel_unique_launch(keys):
    """voxel_unique without the host sync: -> (winners int64 [n] with the m winning point indices first and zeros behind,
    count int32 [1] on the device)."""
    n = keys.shape[0]
    tk, tv, cap = new_table(n, keys.device)
    _lib.call("irx_voxel_insert", _lib.ptr(keys), n, _lib.ptr(tk), _lib.ptr(tv), cap, _stream())
    winners = torch.zeros(max(n, 1), dtype=_i32, device=keys.device)
    count = torch.empty(1, dtype=_i32, device=keys.device)
    _lib.call("irx_voxel_select", _lib.ptr(keys), n, _lib.ptr(tk), _lib.ptr(tv), cap, _lib.ptr(winners),
              _lib.ptr(count), _stream())
    return winners[:n].long(), count


def hash_build(keys):
    n = keys.shape[0]
    tk, tv, cap = new_table(n, keys.device)
    _lib.call("irx_hash_build", _lib.ptr(keys), n, _lib.ptr(tk), _lib.ptr(tv), cap, _stream())
    return tk, tv, cap


def kmap_build_s1(coords, stride, table):
    n = coords.shape[0]
    tk, tv, cap = table
    nbr = torch.empty((27, max(n, 1)), dtype=_i32, device=coords.device)
    _lib.call("irx_kmap_build_s1", _lib.ptr(coords), n, int(stride), _lib.ptr(tk), _lib.ptr(tv), cap,
              _lib.ptr(nbr), max(n, 1), _stream())
    return nbr


def tile_order(tbl, ld, n_out, K):
    """Launch order of the 64-row output tiles of a stride-1 conv over `tbl` (irx_tile_order): int32 [ceil(n_out / 64)]."""
    dev = tbl.device
    nt = (n_out + 63) // 64
    order = torch.empty(max(nt, 1), dtype=_i32, device=dev)
    wsb = int(_lib.load().irx_tile_order_workspace_bytes(n_out))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.call("irx_tile_order", _lib.ptr(tbl), ld, n_out, K, _lib.ptr(order), _lib.ptr(ws), wsb, _stream())
    return order[:nt]


def downsample(keys, coords, stride):
    """-> parent (n,) i32, koff (n,) u8, out_coords (n_out,4), out_keys (n_out,), child (8, ld) i32, ld, n_out."""
    n = coords.shape[0]
    dev = coords.device
    ld = max(n, 1)
    parent = torch.empty(ld, dtype=_i32, device=dev)
    koff = torch.empty(ld, dtype=torch.uint8, device=dev)
    out_coords = torch.empty((ld, 4), dtype=_i32, device=dev)
    out_keys = torch.empty(ld, dtype=_i64, device=dev)
    child = torch.empty((8, ld), dtype=_i32, device=dev)
    n_out = torch.empty(1, dtype=_i32, device=dev)
    wsb = int(_lib.load().irx_downsample_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.call("irx_downsample", _lib.ptr(keys), _lib.ptr(coords), n, int(stride), _lib.ptr(parent),
              _lib.ptr(koff), _lib.ptr(out_coords), _lib.ptr(out_keys), _lib.ptr(child), ld,
              _lib.ptr(n_out), _lib.ptr(ws), wsb, _stream())
    m = int(n_out.item())  # host sync (one per pyramid level)
    return parent[:n], koff[:n], out_coords[:m], out_keys[:m], child, ld, m


class PyramidPending:
    """A pyramid whose kernels are enqueued (irx_pyramid_build) and whose level sizes are on their way to a pinned host
    buffer; finish() waits for that copy (free if enough host work was done in between) and slices the levels.
    n0 is None when the finest level's own row count was still on the device at launch (see pyramid_launch)."""

    def __init__(self, bufs, counts_host, event, n, ld, levels, n0_known):
        self.bufs, self.counts_host, self.event, self.n, self.ld, self.levels = bufs, counts_host, event, n, ld, levels
        self.n0_known = n0_known

    def finish(self):
        """-> (n0, [per-level tuples])"""
        self.event.synchronize()            # the ONE host sync of the pyramid
        m = self.counts_host.tolist()
        if m[0] < 0:
            raise ValueError(RANGE_ERROR)
        n0 = self.n if self.n0_known else m[0]
        m = m[1:]
        parent, koff, out_coords, out_keys, child = self.bufs
        out, n_in = [], n0
        for l in range(self.levels):
            out.append((parent[l, :n_in], koff[l, :n_in], out_coords[l, :m[l]], out_keys[l, :m[l]], child[l], self.ld, m[l]))
            n_in = m[l]
        return n0, out


def pyramid_launch(keys, coords, stride, levels, n0_dev=None):
    """Enqueue `levels` successive down-samplings as ONE library call (irx_pyramid_build: each level reads its row count
    from the device) plus the async D2H copy of the level sizes. -> PyramidPending; all levels share five allocations
    sized for the finest level. n0_dev (int32 device tensor of 1 element): the row count of keys / coords is still on
    the device and their shape[0] is only its upper bound (sync-free voxeliser)."""
    import ctypes
    n = coords.shape[0]
    dev = coords.device
    ld = max(n, 1)
    parent = torch.empty((levels, ld), dtype=_i32, device=dev)
    koff = torch.empty((levels, ld), dtype=torch.uint8, device=dev)
    out_coords = torch.empty((levels, ld, 4), dtype=_i32, device=dev)
    out_keys = torch.empty((levels, ld), dtype=_i64, device=dev)
    child = torch.empty((levels, 8, ld), dtype=_i32, device=dev)
    counts = torch.zeros(levels + 1, dtype=_i32, device=dev)          # [n0, m_1 .. m_levels]
    if n0_dev is not None:
        counts[0:1].copy_(n0_dev)
    wsb = int(_lib.load().irx_downsample_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    arr = ctypes.c_void_p * levels

    def ptrs(t):
        step = t[0].numel() * t.element_size()
        base = t.data_ptr()
        return arr(*[base + l * step for l in range(levels)])

    _lib.call("irx_pyramid_build", _lib.ptr(keys), _lib.ptr(coords), n, int(stride), levels, ptrs(parent), ptrs(koff),
              ptrs(out_coords), ptrs(out_keys), ptrs(child), ld, counts.data_ptr() + 4,
              counts.data_ptr() if n0_dev is not None else None, _lib.ptr(ws), wsb, _stream())
    counts_host = torch.empty(levels + 1, dtype=_i32, pin_memory=True)
    counts_host.copy_(counts, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PyramidPending((parent, koff, out_coords, out_keys, child), counts_host, ev, n, ld, levels, n0_dev is None)


def pyramid_build(keys, coords, stride, levels):
    """-> list of (parent, koff, out_coords, out_keys, child, ld, n_out) per level, the tuples downsample() returns."""
    return pyramid_launch(keys, coords, stride, levels).finish()[1]


def kmap_down_transpose(parent, koff):
    n = parent.shape[0]
    tbl = torch.empty((8, max(n, 1)), dtype=_i32, device=parent.device)
    _lib.call("irx_kmap_down_transpose", _lib.ptr(parent), _lib.ptr(koff), n, _lib.ptr(tbl), max(n, 1),
              _stream())
    return tbl


def bev_table(coords, stride, batch_size, nx, ny, nz, table):
    n = coords.shape[0]
    tk, tv, cap = table
    ncell = batch_size * nx * ny
    tbl = torch.empty((nz, ncell), dtype=_i32, device=coords.device)
    cell = torch.empty(max(n, 1), dtype=_i32, device=coords.device)
    zbin = torch.empty(max(n, 1), dtype=torch.uint8, device=coords.device)
    _lib.call("irx_bev_table", _lib.ptr(coords), n, int(stride), batch_size, nx, ny, nz, _lib.ptr(tk),
              _lib.ptr(tv), cap, _lib.ptr(tbl), ncell, _lib.ptr(cell), _lib.ptr(zbin), _stream())
    return tbl, cell[:n], zbin[:n]


def batch_offsets(coords, nseg):
    off = torch.empty(nseg + 1, dtype=_i32, device=coords.device)
    _lib.call("irx_batch_offsets", _lib.ptr(coords), coords.shape[0], nseg, _lib.ptr(off), _stream())
    return off


# ------------------------------------------------------------------------------ sparse conv ---
def _pairs(tbl, K, n_out):
    """Number of valid (input, output) pairs M of a table (profiling only; cached per table storage). Only the first
    n_out columns are meaningful: tables may be wider (ld > n_out, e.g. the shared pyramid buffers) and the rest is
    uninitialised."""
    if PROFILE_NO_COUNT:
        return 0                   # timeline mode: only the events matter, a count would be a host sync inside the loop
    key = (tbl.data_ptr(), tuple(tbl.shape), K, n_out)
    if key not in _PAIR_COUNT:
        if len(_PAIR_COUNT) > 4096:
            _PAIR_COUNT.clear()
        _PAIR_COUNT[key] = int((tbl[:K, :n_out] >= 0).sum().item())
    return _PAIR_COUNT[key]


def spconv_gather_gemm(x, w, tbl, ld, n_out, K, cin, cout, flip_k, trans_w):
    y = torch.empty((n_out, cout), dtype=_f32, device=x.device)
    wsb = int(_lib.load().irx_spconv_fwd_workspace_bytes(n_out, K, cin, cout, int(trans_w)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    if PROFILE is not None and not PROFILE_NO_COUNT:
        m = _pairs(tbl, K, n_out)
        e0, e1 = _bracket()
    _lib.call("irx_spconv_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(tbl), ld, n_out, K, cin, cout,
              int(flip_k), int(trans_w), _lib.ptr(y), _lib.ptr(ws), wsb, _stream())
    if PROFILE is not None and not PROFILE_NO_COUNT:
        PROFILE.append(("dgrad" if trans_w else "fwd", n_out, K, cin, cout, m, e0, e1))
    return y


def spconv_gather_gemm_t(x, w, tbl, ld, n_out, K, cin, cout, flip_k, trans_w, y_dtype=_f32, accumulate_into=None):
    """irx_spconv_fwd_t: the convolution with typed tensors — x float32 or bfloat16 [n_in][cin], result float32 or bfloat16
    (the encoder executor's bf16 storage mode as a single operator; needs set_compute_dtype('bf16' | 'bf16_operands') for
    bf16 tensors). accumulate_into: y tensor that the result is added to."""
    x_bf = x.dtype == torch.bfloat16
    y = accumulate_into if accumulate_into is not None else torch.empty((n_out, cout), dtype=y_dtype, device=x.device)
    y_bf = y.dtype == torch.bfloat16
    wsb = int(_lib.load().irx_spconv_fwd_workspace_bytes(n_out, K, cin, cout, int(trans_w)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    if PROFILE is not None and not PROFILE_NO_COUNT:
        m = _pairs(tbl, K, n_out)
        e0, e1 = _bracket()
    _lib.call("irx_spconv_fwd_t", _lib.ptr(x), _lib.ptr(w), _lib.ptr(tbl), ld, x.shape[0], n_out, K, cin, cout, int(flip_k),
              int(trans_w), _lib.ptr(y), int(accumulate_into is not None), int(x_bf), int(y_bf), _lib.ptr(ws), wsb, _stream())
    if PROFILE is not None and not PROFILE_NO_COUNT:
        PROFILE.append(("dgrad" if trans_w else "fwd", n_out, K, cin, cout, m, e0, e1))
    return y


def spconv_wgrad(x, dy, tbl, ld, n_out, K, cin, cout):
    dw = torch.empty((K, cin, cout), dtype=_f32, device=x.device)
    wsb = int(_lib.load().irx_spconv_wgrad_workspace_bytes(n_out, K, cin, cout))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    if PROFILE is not None and not PROFILE_NO_COUNT:
        m = _pairs(tbl, K, n_out)
        e0, e1 = _bracket()
    _lib.call("irx_spconv_wgrad", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(tbl), ld, n_out, K, cin, cout,
              _lib.ptr(dw), _lib.ptr(ws), wsb, _stream())
    if PROFILE is not None and not PROFILE_NO_COUNT:
        PROFILE.append(("wgrad", n_out, K, cin, cout, m, e0, e1))
    return dw


def pairs_build(tbl, ld, n_out, K):
    """Compacted per-offset pair lists of a table (device-side, no host sync) -> (in_list, out_list, counts, ldp)."""
    dev = tbl.device
    ldp = max(n_out, 1)
    in_list = torch.empty((K, ldp), dtype=_i32, device=dev)
    out_list = torch.empty((K, ldp), dtype=_i32, device=dev)
    counts = torch.empty(K, dtype=_i32, device=dev)
    wsb = int(_lib.load().irx_pairs_workspace_bytes(n_out, K))
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
    _lib.call("irx_pairs_build", _lib.ptr(tbl), ld, n_out, K, _lib.ptr(in_list), _lib.ptr(out_list), ldp,
              _lib.ptr(counts), _lib.ptr(ws), wsb, _stream())
    return in_list, out_list, counts, ldp


def pairs_build_multi(tables):
    """pairs_build for several tables [(tbl, ld, n_out, K), ...] in ONE library call (two launches) and three
    allocations. -> [(in_list, out_list, counts, ldp), ...]"""
    import ctypes
    nt = len(tables)
    if nt == 0:
        return []
    dev = tables[0][0].device
    lib = _lib.load()
    ldps = [max(n_out, 1) for _, _, n_out, _ in tables]
    sizes = [K * ldp for (_, _, _, K), ldp in zip(tables, ldps)]
    lists = torch.empty(2 * sum(sizes), dtype=_i32, device=dev)
    counts = torch.empty(sum(K for _, _, _, K in tables), dtype=_i32, device=dev)
    wsb = sum(int(lib.irx_pairs_workspace_bytes(n_out, K)) for _, _, n_out, K in tables)
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
    out, il_p, ol_p, cnt_p = [], [], [], []
    lo, co = 0, 0
    for (tbl, ld, n_out, K), ldp, sz in zip(tables, ldps, sizes):
        il = lists[lo:lo + sz].view(K, ldp)
        ol = lists[lo + sz:lo + 2 * sz].view(K, ldp)
        cn = counts[co:co + K]
        lo += 2 * sz
        co += K
        out.append((il, ol, cn, ldp))
        il_p.append(il.data_ptr()); ol_p.append(ol.data_ptr()); cnt_p.append(cn.data_ptr())
    P, I = ctypes.c_void_p * nt, ctypes.c_int * nt
    _lib.call("irx_pairs_build_multi", nt, P(*[t[0].data_ptr() for t in tables]), I(*[t[1] for t in tables]),
              I(*[t[2] for t in tables]), I(*[t[3] for t in tables]), P(*il_p), P(*ol_p), I(*ldps), P(*cnt_p),
              _lib.ptr(ws), wsb, _stream())
    return out


_PAIR_CHANNELS = (32, 64, 128)


def spconv_wgrad_pairs(x, dy, pairs, n_out, K, cin, cout, m_for_profile=None):
    """irx_spconv_wgrad_pairs_t: x, dy both float32 or both bfloat16 (bf16 storage mode; set_compute_dtype('bf16'))"""
    in_list, out_list, counts, ldp = pairs
    dw = torch.empty((K, cin, cout), dtype=_f32, device=x.device)
    wsb = int(_lib.load().irx_spconv_wgrad_pairs_workspace_bytes(n_out, K, cin, cout))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    if PROFILE is not None and not PROFILE_NO_COUNT:
        m = int(counts.sum().item())
        e0, e1 = _bracket()
    bf = x.dtype == torch.bfloat16
    if bf != (dy.dtype == torch.bfloat16):
        raise ValueError("spconv_wgrad_pairs: x and dy must have the same element type")
    _lib.call("irx_spconv_wgrad_pairs_t", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(in_list), _lib.ptr(out_list), ldp,
              _lib.ptr(counts), int(x.shape[0]), n_out, K, cin, cout, _lib.ptr(dw), int(bf), _lib.ptr(ws), wsb, _stream())
    if PROFILE is not None and not PROFILE_NO_COUNT:
        PROFILE.append(("wgrad", n_out, K, cin, cout, m, e0, e1))
    return dw


class SparseConvFn(torch.autograd.Function):
    """y[q] = sum_k x[tbl_f[k][q]] @ w[k].  Backward: data-gradient through `tbl_b` (the same table
    with flipped offsets for a stride-1 conv, the transposed child table for a strided one) and the
    weight-gradient through `tbl_f`."""

    @staticmethod
    def forward(ctx, x, w, tbl_f, ld_f, n_out, tbl_b_fn, flip_b, pairs_fn=None):
        x = _f32c(x)
        w = _f32c(w)
        K, cin, cout = w.shape
        assert x.shape[1] == cin
        y = spconv_gather_gemm(x, w, tbl_f, ld_f, n_out, K, cin, cout, 0, 0)
        ctx.save_for_backward(x, w, tbl_f)
        ctx.meta = (ld_f, n_out, tbl_b_fn, flip_b, pairs_fn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, tbl_f = ctx.saved_tensors
        ld_f, n_out, tbl_b_fn, flip_b, pairs_fn = ctx.meta
        K, cin, cout = w.shape
        dy = _f32c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            tbl_b, ld_b = tbl_b_fn()
            dx = spconv_gather_gemm(dy, w, tbl_b, ld_b, x.shape[0], K, cout, cin, flip_b, 1)
        if ctx.needs_input_grad[1]:
            if pairs_fn is not None and cin in _PAIR_CHANNELS and cout in _PAIR_CHANNELS:
                dw = spconv_wgrad_pairs(x, dy, pairs_fn(), n_out, K, cin, cout)    # dense 64-pair stages
            else:
                dw = spconv_wgrad(x, dy, tbl_f, ld_f, n_out, K, cin, cout)
        return dx, dw, None, None, None, None, None, None


# --------------------------------------------------------------------------------- batchnorm --
def _bn_ws(n, c, device):
    wsb = int(_lib.load().irx_bn_workspace_bytes(n, c))
    return torch.empty(max(wsb, 4), dtype=torch.uint8, device=device), wsb


SYNC_OFF = False   # set by a caller that runs steps on ONE rank of a live group (bench.py's instrumented steps)


def sync_group():
    """The process group sync BatchNorm folds its statistics over: the default group when it has more than one rank."""
    import torch.distributed as dist
    if SYNC_OFF:
        return None
    return dist.group.WORLD if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


class BatchNormActFn(torch.autograd.Function):
    """Train-mode BatchNorm over rows (+ residual) (+ ReLU), one statistics pass + one apply pass.
    sync (a process group): the statistics are taken over the rows of all its ranks (torch.nn.SyncBatchNorm semantics;
    include/irx.h "Sync BatchNorm"): this rank's float64 sums and row count -> all_reduce -> mean / invstd; in the backward
    pass the two gradient sums are folded the same way before the apply pass, and the parameter gradients stay this rank's
    own sums (the gradient all-reduce folds those like every other parameter gradient). Every rank must run the layer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, relu, sync=None):
        x = _f32c(x)
        n, c = x.shape
        dev = x.device
        mean = torch.empty(c, dtype=_f32, device=dev)
        invstd = torch.empty(c, dtype=_f32, device=dev)
        ws, wsb = _bn_ws(n, c, dev)
        res = _f32c(residual) if residual is not None else None
        ctx.sync, ctx.count = sync, None
        if sync is not None:
            import torch.distributed as dist
            sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            _lib.call("irx_bn_sums", _lib.ptr(x), n, c, _lib.ptr(sums), _lib.ptr(ws), wsb, _stream())
            sums[2 * c:].fill_(float(n))
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=sync)
            _lib.call("irx_bn_stats_from_sums", _lib.ptr(sums), 0.0, c, float(eps), float(momentum), _lib.ptr(mean),
                      _lib.ptr(invstd), _lib.ptr(running_mean), _lib.ptr(running_var), _stream())
            ctx.count = sums[2 * c:]                     # folded row count, on the device
        y = torch.empty_like(x)
        g = _f32c(gamma)
        b = _f32c(beta)
        if sync is None:
            # statistics + apply as one call: ONE launch for a tensor that stays on-die (csrc/irx_norm.hip, k_bn_slice_fwd)
            _lib.call("irx_bn_forward", _lib.ptr(x), n, c, float(eps), float(momentum), _lib.ptr(g), _lib.ptr(b), _lib.ptr(res),
                      int(relu), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(y),
                      _lib.ptr(ws), wsb, _stream())
        else:
            _lib.call("irx_bn_apply", _lib.ptr(x), n, c, _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(g),
                      _lib.ptr(b), _lib.ptr(res), int(relu), _lib.ptr(y), _stream())
        ctx.save_for_backward(x, y, mean, invstd, g)
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, g = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=_f32, device=dev)
        dbeta = torch.empty(c, dtype=_f32, device=dev)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        ws, wsb = _bn_ws(n, c, dev)
        if ctx.sync is not None:
            import torch.distributed as dist
            _lib.call("irx_bn_backward_sums", _lib.ptr(x), _lib.ptr(y), _lib.ptr(dy), n, c, _lib.ptr(mean),
                      _lib.ptr(invstd), int(ctx.relu), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), wsb, _stream())
            both = torch.cat([dbeta, dgamma])            # (sum g | sum g xhat) of this rank; folded copies below
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=ctx.sync)
            _lib.call("irx_bn_backward_apply", _lib.ptr(x), _lib.ptr(y), _lib.ptr(dy), n, c, _lib.ptr(mean),
                      _lib.ptr(invstd), _lib.ptr(g), int(ctx.relu), _lib.ptr(both), _lib.ptr(both[c:]), 0.0,
                      _lib.ptr(ctx.count), _lib.ptr(dx), _lib.ptr(dres), _stream())
            return dx, dgamma, dbeta, dres, None, None, None, None, None, None
        _lib.call("irx_bn_backward", _lib.ptr(x), _lib.ptr(y), _lib.ptr(dy), n, c, _lib.ptr(mean),
                  _lib.ptr(invstd), _lib.ptr(g), int(ctx.relu), _lib.ptr(dx), _lib.ptr(dgamma),
                  _lib.ptr(dbeta), _lib.ptr(dres), _lib.ptr(ws), wsb, _stream())
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None


def bn_eval(x, gamma, beta, residual, running_mean, running_var, eps, relu):
    """Inference-mode BN(+res)(+ReLU) with running statistics: the apply kernel, autograd through torch."""
    invstd = torch.rsqrt(running_var.float() + eps)
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad):
        y = (x - running_mean) * (invstd * gamma) + beta
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    x = _f32c(x)
    n, c = x.shape
    y = torch.empty_like(x)
    res = _f32c(residual) if residual is not None else None
    _lib.call("irx_bn_apply", _lib.ptr(x), n, c, _lib.ptr(_f32c(running_mean)), _lib.ptr(invstd),
              _lib.ptr(_f32c(gamma)), _lib.ptr(_f32c(beta)), _lib.ptr(res), int(relu), _lib.ptr(y), _stream())
    return y


# ------------------------------------------------------------------------- segmented max -------
class SegmentMaxFn(torch.autograd.Function):
    """Channel-wise max over contiguous row segments; gradient goes to the arg-max row."""

    @staticmethod
    def forward(ctx, x, offsets, nseg):
        x = _f32c(x)
        c = x.shape[1]
        y = torch.empty((nseg, c), dtype=_f32, device=x.device)
        arg = torch.empty((nseg, c), dtype=_i32, device=x.device)
        _lib.call("irx_segment_max", _lib.ptr(x), _lib.ptr(offsets), nseg, c, _lib.ptr(y), _lib.ptr(arg),
                  _stream())
        ctx.save_for_backward(arg)
        ctx.shape = tuple(x.shape)
        ctx.mark_non_differentiable(arg)
        return y, arg

    @staticmethod
    def backward(ctx, dy, _darg):
        (arg,) = ctx.saved_tensors
        n, c = ctx.shape
        dx = torch.zeros((n, c), dtype=_f32, device=dy.device)
        dy = _f32c(dy)
        _lib.call("irx_segment_max_backward", _lib.ptr(dy), _lib.ptr(arg), arg.shape[0], c, _lib.ptr(dx),
                  _stream())
        return dx, None, None


def segment_max(x, offsets, nseg):
    return SegmentMaxFn.apply(x, offsets, nseg)[0]


def segment_mean(x):
    """x (nseg, len, c) f32 cuda -> (nseg, c)."""
    x = _f32c(x)
    nseg, ln, c = x.shape
    y = torch.empty((nseg, c), dtype=_f32, device=x.device)
    _lib.call("irx_segment_mean", _lib.ptr(x), nseg, ln, c, _lib.ptr(y), _stream())
    return y


def knn_batched(sup_xyz, sup_offsets, qry_xyz, qry_batch, k):
    sup_xyz = _f32c(sup_xyz)
    qry_xyz = _f32c(qry_xyz)
    nq = qry_xyz.shape[0]
    out = torch.empty((nq, k), dtype=_i32, device=sup_xyz.device)
    _lib.call("irx_knn_batched", _lib.ptr(sup_xyz), _lib.ptr(sup_offsets), _lib.ptr(qry_xyz),
              _lib.ptr(qry_batch), nq, k, _lib.ptr(out), _stream())
    return out
