"""Dev tool: the stem convolution forward (3^3, 7 -> 32 channels) alone on a pyramid level of the bench's size, both kernels
(IRX_STEM_MFMA=0 / 1 select them per process), and a value check of one against the other is done by the caller's tests."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.sparse import functional as F_
from instancerefer_amd.sparse.tensor import SparseTensor
lib = _lib.load(); dev = torch.device("cuda")
res = S.to_device(S.make_batch(16, seed=123), dev)
st = res["lidar"].canonical() if hasattr(res["lidar"], "canonical") else res["lidar"]
lv = st.level()
tbl, ld = lv.nbr27()
n = lv.n
x = st.F.contiguous().float()
w = torch.randn(27, 7, 32, device=dev) * 0.1
y = torch.empty(n, 32, device=dev)
wsb = lib.irx_spconv_fwd_workspace_bytes(n, 27, 7, 32, 0); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
def run():
    _lib.call("irx_spconv_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(tbl), ld, n, 27, 7, 32, 0, 0, _lib.ptr(y), _lib.ptr(ws), wsb, _lib.stream_ptr())
outs = {}
for flag in (0, 1):
    _lib.set_knob("stem_mfma", flag)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    outs[flag] = y.clone()
    print("stem_mfma=%d: stem forward on %d voxels: %.1f us" % (flag, n, e0.elapsed_time(e1) / 50 * 1e3))
d = (outs[0] - outs[1]).abs()
print("max abs difference between the two kernels: %.3e (max |y| %.3f); rows differing > 1e-4: %d" % (float(d.max()), float(outs[0].abs().max()), int((d.max(1).values > 1e-4).sum())))
bad = torch.nonzero(d.max(1).values > 1e-4).flatten()[:8].tolist()
print("first differing rows:", bad)
