"""Dev tool: time irx_spconv_fwd on synthetic scene levels (events), e.g. under IRX_SPCONV_DBG masks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import synthetic as S
from instancerefer_amd.sparse import functional as F_
from instancerefer_amd.sparse.utils import voxelize
dev = torch.device('cuda')
import instancerefer_amd as irx
if os.environ.get('IRX_DTYPE'): irx.set_compute_dtype(os.environ['IRX_DTYPE'])
print('compute dtype', irx.get_compute_dtype())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dd = S.make_batch(B, seed=123)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, B)
lv = st.level()
levels = []
for s in range(5):
    levels.append(lv)
    if s < 4: lv = lv.down().out_level
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for li, cin, cout in ((1, 64, 64), (2, 128, 128), (3, 128, 128), (4, 128, 128)):
    lv = levels[li]; n = lv.n
    tbl, ld = lv.nbr27()
    M = int((tbl >= 0).sum())
    x = torch.randn(n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.05
    us = bench(lambda: F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 1, 1))
    dy = torch.randn(n, cout, device=dev)
    pairs = lv.pairs27()
    usw = bench(lambda: F_.spconv_wgrad_pairs(x, dy, pairs, n, 27, cin, cout))
    print('stride %2d n=%7d M=%8d %3d->%3d  dgrad %7.1f us %6.2f TF | wgrad(pairs) %7.1f us %6.2f TF' % (lv.stride, n, M, cin, cout, us, 2.0 * M * cin * cout / us / 1e6, usw, 2.0 * M * cin * cout / usw / 1e6))
