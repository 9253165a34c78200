import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import synthetic as S
from instancerefer_amd.sparse import functional as F_
from instancerefer_amd.sparse.utils import voxelize
dev = torch.device('cuda')
dd = S.make_batch(3, seed=123, num_points=6000, num_instances=6, num_candidates=[4, 1, 3], points_per_instance=256)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 3)
lv = st.level().down().out_level.down().out_level
g = torch.Generator(device='cpu').manual_seed(1)
n = lv.n
for cin, cout in ((128, 128), (64, 64)):
    x = torch.randn(n, cin, generator=g).to(dev); w = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev)
    tbl, ld = lv.nbr27()
    y = F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 0, 0)
    ref = torch.zeros(n, cout, dtype=torch.float64, device=dev)
    xd, wd = x.double(), w.double()
    for k in range(27):
        idx = tbl[k, :n].long(); val = idx >= 0
        ref[val] += xd[idx[val]] @ wd[k]
    err = (y.double() - ref).abs()
    print(os.environ.get('IRX_SPCONV_V1', 'v2'), cin, cout, 'max|ref| %.3f maxerr %.3e  rows with err>1e-5: %d  worst row %d' % (
        ref.abs().max().item(), err.max().item(), int((err.max(1)[0] > 1e-5).sum()), int(err.max(1)[0].argmax())))
def check(name, x, w, tbl, ld, n_out, K):
    cin, cout = w.shape[1], w.shape[2]
    y = F_.spconv_gather_gemm(x, w, tbl, ld, n_out, K, cin, cout, 0, 0)
    ref = torch.zeros(n_out, cout, dtype=torch.float64, device=dev)
    xd, wd = x.double(), w.double()
    for k in range(K):
        idx = tbl[k, :n_out].long(); val = idx >= 0
        ref[val] += xd[idx[val]] @ wd[k]
    err = (y.double() - ref).abs()
    rowerr = err.max(1)[0]
    print(os.environ.get('IRX_SPCONV_V1', 'v2'), name, 'max|ref| %.3f maxerr %.3e  rows err>2e-6*max: %d / %d  nan %d' % (
        ref.abs().max().item(), err.max().item(), int((rowerr > 2e-6 * ref.abs().max()).sum()), n_out, int(torch.isnan(y).sum())))
lv0 = st.level()
lvl = lv0
for name, cin, cout in (('down1 32->64', 32, 64), ('down2 64->128', 64, 128), ('down3 128->128', 128, 128), ('down4 128->128', 128, 128)):
    dm = lvl.down(); o = dm.out_level
    x = torch.randn(lvl.n, cin, generator=g).to(dev); w = (torch.randn(8, cin, cout, generator=g) * 0.1).to(dev)
    check(name, x, w, dm.child, dm.ld, o.n, 8)
    lvl = o
tblb, cell, zbin = lvl.bev(15, 25, 5)
x = torch.randn(lvl.n, 128, generator=g).to(dev); w = (torch.randn(5, 128, 128, generator=g) * 0.1).to(dev)
check('bev', x, w, tblb, 3 * 375, 3 * 375, 5)
