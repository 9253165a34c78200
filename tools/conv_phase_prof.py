"""Dev tool: per-phase cycle attribution of k_spconv2 on the stride-4 128->128 layer.

Build the instrumented variant first (it is not part of the product library):
  cd instancerefer_amd/csrc && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DIRX_S2_PROF -c $f -o /tmp/prof_$f.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/prof_*.o -o tools/micro/libirx_prof.so
then run with IRX_LIB_PATH=tools/micro/libirx_prof.so.
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import synthetic as S, _lib
from instancerefer_amd.sparse import functional as F_
from instancerefer_amd.sparse.utils import voxelize
dev = torch.device('cuda')
dd = S.make_batch(16, seed=123)
pts = [torch.from_numpy(p) for p in dd['scene_points']]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 16)
lv = st.level()
cin = cout = int(os.environ.get('CH', '128'))
for s in range(2 if cin == 128 else 1): lv = lv.down().out_level
n = lv.n; tbl, ld = lv.nbr27()
x = torch.randn(n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.05
lib = _lib.load()
import instancerefer_amd as irx
if os.environ.get('IRX_DTYPE'): irx.set_compute_dtype(os.environ['IRX_DTYPE'])
print('compute dtype', irx.get_compute_dtype())
lib.irx_debug_s2_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 1, 1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 1, 1)
e1.record(); torch.cuda.synchronize()
pairs = int((tbl[:27 * ld].view(27, ld)[:, :n] >= 0).sum())
us = e0.elapsed_time(e1) * 100.0
print('launch (incl. weight permute) %.1f us, pairs %d (%.1f per row), %.1f TFLOP/s' % (us, pairs, pairs / n, 2.0 * pairs * cin * cout / us / 1e6))
lib.irx_debug_s2_prof(None, 1)
F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 1, 1); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lib.irx_debug_s2_prof(out, 0)
v = list(out)
names = ['loop-top', 'barrier1', 'vmcnt(0)', 'sA write', 'barrier2', 'lookahead', 'mfma groups']
waves = v[9]
print('n=%d %d->%d waves %d, avg cycles per wave (main loop): %.0f' % (n, cin, cout, waves, v[8] / waves))
for i, nm in enumerate(names): print('  %-16s %9.0f cycles/wave  %5.1f%%' % (nm, v[i] / waves, 100.0 * v[i] / v[8]))
