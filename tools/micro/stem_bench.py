"""Dev: time the stem convolution (7 -> 32, K = 27) forward and weight gradient on the stride-1 level of 16 scenes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instancerefer_amd import synthetic as S
from instancerefer_amd.sparse import functional as F_
from instancerefer_amd.sparse.utils import voxelize
dev = torch.device('cuda')
dd = S.make_batch(16, seed=123)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 16)
lv = st.level(); n = lv.n; tbl, ld = lv.nbr27()
x = torch.randn(n, 7, device=dev); w = torch.randn(27, 7, 32, device=dev) * 0.1
def bench(fn, r=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(r): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / r * 1e3
y = F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, 7, 32, 0, 0)
print('n=%d stem fwd %.1f us  checksum %.6f' % (n, bench(lambda: F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, 7, 32, 0, 0)), float(y.double().sum())))
