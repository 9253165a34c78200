"""Dev tool: HBM throughput of the BatchNorm kernels on encoder-sized tensors."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import _lib
from instancerefer_amd.sparse import functional as F_
lib = _lib.load(); dev = torch.device('cuda')
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n, c in ((488800, 32), (258865, 64), (81261, 128), (20267, 128), (4636, 128)):
    x = torch.randn(n, c, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x); dres = torch.empty_like(x)
    mean = torch.zeros(c, device=dev); inv = torch.ones(c, device=dev); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
    wsb = lib.irx_bn_workspace_bytes(n, c); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    s = _lib.stream_ptr()
    t_stats = bench(lambda: lib.irx_bn_stats(x.data_ptr(), n, c, 1e-5, 0.1, mean.data_ptr(), inv.data_ptr(), None, None, ws.data_ptr(), wsb, s))
    t_apply = bench(lambda: lib.irx_bn_apply(x.data_ptr(), n, c, mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), None, 1, y.data_ptr(), s))
    t_applyr = bench(lambda: lib.irx_bn_apply(x.data_ptr(), n, c, mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), dy.data_ptr(), 1, y.data_ptr(), s))
    t_bwd = bench(lambda: lib.irx_bn_backward(x.data_ptr(), y.data_ptr(), dy.data_ptr(), n, c, mean.data_ptr(), inv.data_ptr(), g.data_ptr(), 1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), wsb, s))
    t_bwdr = bench(lambda: lib.irx_bn_backward(x.data_ptr(), y.data_ptr(), dy.data_ptr(), n, c, mean.data_ptr(), inv.data_ptr(), g.data_ptr(), 1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), dres.data_ptr(), ws.data_ptr(), wsb, s))
    mb = n * c * 4 / 1e6
    print('n=%7d c=%3d (%6.1f MB): stats %6.1f us (%5.0f GB/s) | apply %6.1f us (%5.0f GB/s) | apply+res %6.1f (%5.0f) | bwd %6.1f us (%5.0f GB/s, 7 passes) | bwd+res %6.1f (%5.0f, 8 passes)' % (
        n, c, mb, t_stats, mb / t_stats * 1e3, t_apply, 2 * mb / t_apply * 1e3, t_applyr, 3 * mb / t_applyr * 1e3, t_bwd, 7 * mb / t_bwd * 1e3, t_bwdr, 8 * mb / t_bwdr * 1e3))
