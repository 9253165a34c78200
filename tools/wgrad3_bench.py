"""Dev tool: the bf16-row pair-list weight gradient as one operator (irx_spconv_wgrad_pairs_t) on the levels of the bench
scene pyramid — k_wgrad3 (knob "wgrad3" = 1) against the widening kernel (0), both checked against a float64 torch evaluation
on the same bf16 values. Usage: python tools/wgrad3_bench.py [B] [reps]"""
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


def ref_wgrad(x, dy, tbl, n_out, K):
    out = []
    for k in range(K):
        idx = tbl[k, :n_out].long()
        v = idx >= 0
        out.append(x[idx[v]].double().t() @ dy[:n_out][v].double())
    return torch.stack(out)


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
    return e0.elapsed_time(e1) / REPS * 1e3


def case(name, tbl, ld, n_in, n_out, K, cin, cout):
    if ONLY and not any(t in name for t in ONLY.split(',')):
        return
    g = torch.Generator(device=dev).manual_seed(n_out + K + cin)
    x = torch.randn(n_in, cin, device=dev, generator=g).bfloat16()
    dy = torch.randn(n_out, cout, device=dev, generator=g).bfloat16()
    pairs = F_.pairs_build(tbl, ld, n_out, K)
    m = int(pairs[2].sum())
    res = {}
    r = ref_wgrad(x, dy, tbl, n_out, K) if CHECK else None
    for v3 in (0, 1):
        _lib.set_knob('wgrad3', v3)
        run = lambda: F_.spconv_wgrad_pairs(x, dy, pairs, n_out, K, cin, cout)
        dw = run()
        us = timed(run)
        err = float((dw.double() - r).abs().max() / r.abs().max()) if CHECK else float('nan')
        res[v3] = (us, err)
    print('%-18s n_out %7d K %2d %3d->%3d M %8d | old %7.1f us (err %.1e) | k_wgrad3 %7.1f us (err %.1e) | x%.2f | %.0f TFLOP/s useful'
          % (name, n_out, K, cin, cout, m, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] / res[1][0],
             2.0 * m * cin * cout / res[1][0] / 1e6))


lv = st.level()
levels = [lv]
for s in range(4):
    lv = lv.down().out_level
    levels.append(lv)
chans = [32, 64, 128, 128, 128]
for i, lv in enumerate(levels):
    c = chans[i]
    tbl, ld = lv.nbr27()
    if c >= 64:
        case('stride %d 3^3' % lv.stride, tbl, ld, lv.n, lv.n, 27, c, c)
    if i < 4:
        dm = lv.down()
        co = chans[i + 1]
        if c >= 64:
            case('down %d->%d' % (lv.stride, lv.stride * 2), dm.child, dm.ld, lv.n, dm.out_level.n, 8, c, co)
