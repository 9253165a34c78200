"""Dev tool: the bf16-input sparse conv as one operator (irx_spconv_fwd_t) on the levels of the bench scene pyramid —
third-generation kernel (irx_spconv3.hip) against the second (knob "spconv3" = 0), both checked against a float64 torch
evaluation on the bf16-rounded operands. Usage: python tools/conv3_bench.py [B] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instancerefer_amd as irx
from instancerefer_amd import synthetic as S, _lib
from instancerefer_amd.sparse.utils import voxelize
from instancerefer_amd.sparse import functional as F_

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CHECK = os.environ.get('CHECK', '1') != '0'
ONLY = os.environ.get('ONLY', '')
dd = S.make_batch(B, seed=123)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, B)
irx.set_compute_dtype('bf16')


def ref_conv(x, w, tbl, n_out, K, flip, trans):
    """float64 gather-GEMM on bf16-rounded operands (only rounding differences of the fp32 sums remain)"""
    wb = w.bfloat16().double()
    y = torch.zeros(n_out, (w.shape[1] if trans else w.shape[2]), dtype=torch.float64, device=dev)
    for k in range(K):
        kt = K - 1 - k if flip else k
        idx = tbl[kt, :n_out].long()
        v = idx >= 0
        wk = wb[k].t() if trans else wb[k]
        y[v] += x[idx[v]].double() @ wk
    return y


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / REPS * 1e3
    # the conv kernel alone (events recorded by the library right around it: no weight-image / split-reduce launches)
    ks = []
    for _ in range(min(REPS, 10)):
        b0, b1 = F_._bracket()
        fn()
        torch.cuda.synchronize()
        ks.append(b0.elapsed_time(b1) * 1e3)
    ks.sort()
    timed.kernel_us = ks[len(ks) // 2]
    return tot


def case(name, tbl, ld, n_in, n_out, K, cin, cout, flip, trans):
    if ONLY and not any(t in '%s %d->%d' % (name, cin, cout) for t in ONLY.split(',')):
        return
    g = torch.Generator(device=dev).manual_seed(n_out + K + cin)
    x = torch.randn(n_in, cin, device=dev, generator=g).bfloat16()
    w = (torch.randn(K, cout, cin, device=dev, generator=g) if trans else torch.randn(K, cin, cout, device=dev, generator=g)) * 0.05
    m = int((tbl[:K, :n_out] >= 0).sum())
    res = {}
    for v3 in (0, 1):
        _lib.set_knob('spconv3', v3)
        run = lambda: F_.spconv_gather_gemm_t(x, w, tbl, ld, n_out, K, cin, cout, flip, trans, y_dtype=torch.bfloat16)
        y = run()
        us = timed(run)
        err = float('nan')
        if CHECK:
            r = ref_conv(x, w, tbl, n_out, K, flip, trans)
            yf = F_.spconv_gather_gemm_t(x, w, tbl, ld, n_out, K, cin, cout, flip, trans, y_dtype=torch.float32)
            err = float((yf.double() - r).abs().max() / r.abs().max())
            # accumulate path: y0 + result
            y0 = torch.randn(n_out, cout, device=dev).bfloat16()
            ya = F_.spconv_gather_gemm_t(x, w, tbl, ld, n_out, K, cin, cout, flip, trans, accumulate_into=y0.clone())
            ea = float((ya.double() - (y0.double() + r).bfloat16().double()).abs().max() / r.abs().max())
            err = max(err, ea / 4)   # (bf16 rounding of the sum: one ulp of slack)
        res[v3] = (us, err, timed.kernel_us)
    algo = 2 * (m * cin + n_out * cout + K * cin * cout) + 8 * m
    print('%-22s n_out %7d K %2d %3d->%3d M %8d | v2 %7.1f us (err %.1e) | v3 %7.1f us (err %.1e) | x%.2f | kernel only v2 %6.1f v3 %6.1f us | v3 %.0f GB/s algo (kernel), %.0f TFLOP/s useful'
          % (name, n_out, K, cin, cout, m, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] / res[1][0], res[0][2], res[1][2],
             algo / res[1][2] / 1e3, 2.0 * m * cin * cout / res[1][2] / 1e6))


lv = st.level()
levels = [lv]
for s in range(4):
    lv = lv.down().out_level
    levels.append(lv)
chans = [32, 64, 128, 128, 128]
for i, lv in enumerate(levels):
    c = chans[i]
    tbl, ld = lv.nbr27()
    if c * c >= 2048:
        case('stride %d 3^3 fwd' % lv.stride, tbl, ld, lv.n, lv.n, 27, c, c, 0, 0)
        case('stride %d 3^3 dgrad' % lv.stride, tbl, ld, lv.n, lv.n, 27, c, c, 1, 1)
    if i < 4:
        dm = lv.down()
        co = chans[i + 1]
        no = dm.out_level.n
        case('down %d->%d fwd' % (lv.stride, lv.stride * 2), dm.child, dm.ld, lv.n, no, 8, c, co, 0, 0)
        tb = F_.kmap_down_transpose(dm.parent, dm.koff)
        case('down %d->%d dgrad' % (lv.stride, lv.stride * 2), tb, max(lv.n, 1), no, lv.n, 8, co, c, 0, 1)
