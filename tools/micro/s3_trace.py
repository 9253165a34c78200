"""Dev: per-wave s_memtime stamps of k_spconv3 (library built with -DIRX_S3_TRACE=1, IRX_LIB_PATH=tools/micro/libirx_s3_trace.so).
Usage: IRX_LIB_PATH=... python tools/micro/s3_trace.py [B] [level stride list, e.g. 4,8,16]"""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import instancerefer_amd as irx
from instancerefer_amd import synthetic as S, _lib
from instancerefer_amd.sparse.utils import voxelize
from instancerefer_amd.sparse import functional as F_

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
WANT = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else '4,8,16').split(',')]
dd = S.make_batch(B, seed=123)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, B)
irx.set_compute_dtype('bf16')
lib = _lib.load()
lib.irx_debug_s3_trace.argtypes = [ctypes.c_void_p]
lib.irx_debug_s3_trace.restype = ctypes.c_int

lv = st.level()
levels = [lv]
for s in range(4):
    lv = lv.down().out_level
    levels.append(lv)
chans = [32, 64, 128, 128, 128]
NW = 4
for i, lv in enumerate(levels):
    if lv.stride not in WANT:
        continue
    c = chans[i]
    tbl, ld = lv.nbr27()
    n = lv.n
    g = torch.Generator(device=dev).manual_seed(n)
    x = torch.randn(n, c, device=dev, generator=g).bfloat16()
    w = torch.randn(27, c, c, device=dev, generator=g) * 0.05
    run = lambda: F_.spconv_gather_gemm_t(x, w, tbl, ld, n, 27, c, c, 0, 0, y_dtype=torch.bfloat16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    nblk = 8 * ((n + 127) // 128 + 8) * 9 + 64
    buf = torch.zeros(nblk * NW * 256, dtype=torch.int64, device=dev)
    assert lib.irx_debug_s3_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    lib.irx_debug_s3_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 256)
    t = t[t[:, 0] != 0]
    nit = t[:, 253].astype(int)
    t0 = t[:, 0].min()
    hw = t[:, 255]
    # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... ; xcc via different reg
    cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 12) & 1) << 7)
    print('== stride %d  n %d  c %d: %d waves recorded, items per wave min/mean/max %d / %.1f / %d' % (lv.stride, n, c, len(t), nit.min(), nit.mean(), nit.max()))
    clk = 1.0   # cycles (s_memtime ticks at the shader clock per the guide)
    tot = (t[:, 251] - t[:, 0])
    print('   wave lifetime  mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f cycles' % (tot.mean(), *np.percentile(tot, [10, 50, 90]), tot.max()))
    print('   start offsets  p50 %.0f  p90 %.0f  max %.0f ; end offsets p10 %.0f p50 %.0f max %.0f' % (
        *np.percentile(t[:, 0] - t0, [50, 90]), (t[:, 0] - t0).max(), *np.percentile(t[:, 251] - t0, [10, 50]), (t[:, 251] - t0).max()))
    print('   setup: table->sOff %.0f | masks %.0f | prologue loads %.0f | epilogue %.0f' % (
        (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 251] - t[:, 250]).mean()))
    w_, c_, b_, n_ = [], [], [], 0
    for r in range(len(t)):
        k = nit[r]
        ev = t[r, 4:4 + 4 * k].reshape(k, 4)
        w_.append(ev[:, 1] - ev[:, 0]); c_.append(ev[:, 2] - ev[:, 1]); b_.append(ev[:, 3] - ev[:, 2])
    w_, c_, b_ = np.concatenate(w_), np.concatenate(c_), np.concatenate(b_)
    print('   per item: wait(step 0) mean %.0f p50 %.0f p90 %.0f | steps mean %.0f p50 %.0f p90 %.0f | barrier mean %.0f p50 %.0f p90 %.0f | total %.0f' % (
        w_.mean(), *np.percentile(w_, [50, 90]), c_.mean(), *np.percentile(c_, [50, 90]), b_.mean(), *np.percentile(b_, [50, 90]), (w_ + c_ + b_).mean()))
    loop = t[:, 250] - t[:, 3]
    print('   loop total mean %.0f cycles; lifetime split: setup %.1f %% loop %.1f %% epilogue %.1f %%' % (
        loop.mean(), 100 * (t[:, 3] - t[:, 0]).mean() / tot.mean(), 100 * loop.mean() / tot.mean(), 100 * (t[:, 251] - t[:, 250]).mean() / tot.mean()))
    # steps time split by live / skipped (steps < 200 cycles = skipped)
    sk = c_ < 150
    print('   items skipped (steps < 150 cyc): %.1f %%; live steps mean %.0f; skipped mean %.0f' % (100 * sk.mean(), c_[~sk].mean(), c_[sk].mean() if sk.any() else 0))
    ucu, cnt = np.unique(cu, return_counts=True)
    print('   distinct cu ids %d (waves per id min/mean/max %d / %.1f / %d)' % (len(ucu), cnt.min(), cnt.mean(), cnt.max()))
    print('   span first start -> last end: %.0f cycles' % (t[:, 251].max() - t0))
