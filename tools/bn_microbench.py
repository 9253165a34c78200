"""Dev tool: the BatchNorm operators in isolation on encoder-sized tensors (fp32 and bf16 storage): microseconds per call and the
algorithmic HBM rate (forward 3 passes of N*C*e, backward with ReLU 6 passes — x, y | mask, dy read twice... counted
as bytes actually named by the formulation: fwd 3, bwd 5 without a shortcut (remask) / 7 with one).
  python tools/bn_microbench.py            (env IRX_BN_SLICE_BYTES / IRX_BN_LASTBLOCK select the paths)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import _lib
lib = _lib.load(); dev = torch.device('cuda')
def bench(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SIZES = ((488800, 32), (258865, 64), (81261, 128), (61457, 32), (51417, 64), (29101, 128), (20267, 128), (8990, 128), (4636, 128), (2244, 128), (6000, 128))
print("slice<=%s lastblock=%s" % (os.environ.get("IRX_BN_SLICE_BYTES", "default"), os.environ.get("IRX_BN_LASTBLOCK", "1")))
for bf in (1, 0):
    for n, c in SIZES:
        dt = torch.bfloat16 if bf else torch.float32
        x = torch.randn(n, c, device=dev).to(dt); y = torch.empty_like(x); dy = torch.randn(n, c, device=dev).to(dt); dx = torch.empty_like(x)
        dres = torch.empty_like(x); res = torch.randn(n, c, device=dev).to(dt)
        mean = torch.zeros(c, device=dev); inv = torch.ones(c, device=dev); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
        dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
        wsb = lib.irx_bn_workspace_bytes(n, c); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        s = _lib.stream_ptr()
        P = lambda t: t.data_ptr()
        t_f = bench(lambda: lib.irx_bn_forward_ex(P(x), n, c, 1e-5, 0.1, P(g), P(b), None, 1, P(mean), P(inv), None, None, P(y), P(ws), wsb, s, bf, bf, bf))
        t_fr = bench(lambda: lib.irx_bn_forward_ex(P(x), n, c, 1e-5, 0.1, P(g), P(b), P(res), 1, P(mean), P(inv), None, None, P(y), P(ws), wsb, s, bf, bf, bf))
        t_b = bench(lambda: lib.irx_bn_backward_ex(P(x), P(y), P(dy), n, c, P(mean), P(inv), P(g), P(b), 1, P(dx), P(dg), P(db), None, P(ws), wsb, s, bf, bf, bf, bf, bf))
        t_br = bench(lambda: lib.irx_bn_backward_ex(P(x), P(y), P(dy), n, c, P(mean), P(inv), P(g), None, 1, P(dx), P(dg), P(db), P(dres), P(ws), wsb, s, bf, bf, bf, bf, bf))
        mb = n * c * (2 if bf else 4) / 1e6
        print('%s n=%7d c=%3d (%5.1f MB): fwd %6.1f us (%5.0f GB/s) | fwd+res %6.1f (%5.0f) | bwd %6.1f (%5.0f) | bwd+res %6.1f (%5.0f)' % (
            "bf16" if bf else "f32 ", n, c, mb, t_f, 3 * mb / t_f * 1e3, t_fr, 4 * mb / t_fr * 1e3, t_b, 5 * mb / t_b * 1e3, t_br, 8 * mb / t_br * 1e3))
