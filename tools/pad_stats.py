"""Dev tool: padding statistics of the per-(tile, offset) pair compaction (rows executed by the MFMA / useful pairs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import synthetic as S
from instancerefer_amd.sparse.utils import voxelize
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dd = S.make_batch(B, seed=123)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, B)
lv = st.level()
for s in range(5):
    tbl, ld = lv.nbr27()
    n = lv.n
    valid = (tbl[:, :n] >= 0)
    M = int(valid.sum())
    line = 'stride %2d n=%7d M=%8d pairs/row %.2f |' % (lv.stride, n, M, M / n)
    for tm in (16, 32, 64, 128):
        npad = (n + tm - 1) // tm * tm
        v = torch.zeros(27, npad, dtype=torch.bool, device=dev); v[:, :n] = valid
        c = v.view(27, npad // tm, tm).sum(-1)
        for gran in (4, 16):
            padded = ((c + gran - 1) // gran * gran).sum().item()
            line += ' TM%d/g%d %.3f' % (tm, gran, padded / M)
        if tm == 64:
            hist = torch.bincount(((c + 15) // 16).flatten(), minlength=5).tolist()
            line += ' groups-hist %s' % hist
    print(line)
    if s < 4: lv = lv.down().out_level
