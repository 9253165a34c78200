"""Dev: k_spconv2t (128 rows x 64 columns per workgroup) against k_spconv2 on the synthetic scene levels: same outputs
up to fp32 summation order, timing of both. Runs itself twice with IRX_SPCONV_TALL=0 / 1."""
import os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.sparse import functional as F_
    from instancerefer_amd.sparse.utils import voxelize
    dev = torch.device('cuda')
    dd = S.make_batch(16, seed=123)
    pts = [torch.from_numpy(p) for p in dd['scene_points']]
    allp = torch.cat(pts).to(dev)
    batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
    st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 16)
    lv = st.level(); levels = []
    for s in range(5):
        levels.append(lv)
        if s < 4: lv = lv.down().out_level
    out = {}
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    for li, cin, cout in ((1, 64, 64), (2, 64, 128), (2, 128, 128), (2, 128, 64), (3, 128, 128), (4, 128, 128)):
        lv = levels[li]; n = lv.n
        tbl, ld = lv.nbr27()
        x = torch.randn(n, cin, device=dev, generator=g); w = torch.randn(27, cin, cout, device=dev, generator=g) * 0.05
        for flip in (0, 1):
            y = F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, flip, flip)
            out[(li, cin, cout, flip)] = y.cpu()
        for _ in range(3): F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 1, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 1, 1)
        e1.record(); torch.cuda.synchronize()
        M = int((tbl >= 0).sum())
        us = e0.elapsed_time(e1) * 50.0
        print('  n=%7d %3d->%3d  %7.1f us %6.2f TF' % (n, cin, cout, us, 2.0 * M * cin * cout / us / 1e6), flush=True)
    torch.save(out, sys.argv[1])
else:
    res = []
    for tall in ('0', '1'):
        f = '/tmp/tall_%s.pt' % tall
        print('IRX_SPCONV_TALL=' + tall, flush=True)
        subprocess.run([sys.executable, __file__, f], env=dict(os.environ, IRX_SPCONV_TALL=tall), check=True)
        res.append(torch.load(f))
    for key in res[0]:
        a, b = res[0][key], res[1][key]
        print(key, 'max|diff| %.3g  rel %.3g' % ((a - b).abs().max(), (a - b).abs().max() / a.abs().max()))
