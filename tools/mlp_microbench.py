"""Dev tool: the seven head MLPs of the model at B = 16 / Nc = 64 through irx_mlp2_fwd / irx_mlp2_bwd in isolation: us per call
(forward = 2 launches, backward = 2-3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import _lib
lib = _lib.load(); dev = torch.device('cuda')
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (name, rows, din, dh, dout, norm 1 = BatchNorm train / 3 = LayerNorm)
SHAPES = (("attribute.lang_emb_fc", 16, 256, 256, 256, 1), ("attribute.vis_emb_fc", 64, 128, 256, 256, 3),
          ("relation.lang_emb_fc", 16, 256, 128, 128, 1), ("relation.vis_emb_fc", 64, 128, 128, 128, 3),
          ("scene.lang_emb_fc", 16, 256, 128, 128, 3), ("scene.vis_emb_fc1", 64, 128, 128, 128, 3), ("scene.cls", 16, 128, 128, 9, 1),
          ("B=128: attribute.vis_emb_fc", 512, 128, 256, 256, 3))
P = lambda t: t.data_ptr()
tf = tb = 0.0
for name, rows, din, dh, dout, norm in SHAPES:
    x = torch.randn(rows, din, device=dev); w1 = torch.randn(dh, din, device=dev) * 0.05; b1 = torch.zeros(dh, device=dev)
    g = torch.ones(dh, device=dev); be = torch.zeros(dh, device=dev); w2 = torch.randn(dout, dh, device=dev) * 0.05; b2 = torch.zeros(dout, device=dev)
    rm = torch.zeros(dh, device=dev); rv = torch.ones(dh, device=dev)
    saved = torch.empty(lib.irx_mlp2_saved_floats(rows, dh), device=dev); y = torch.empty(rows, dout, device=dev); dy = torch.randn(rows, dout, device=dev)
    dhid = torch.empty(rows, dh, device=dev); dx = torch.empty(rows, din, device=dev)
    dw1 = torch.empty_like(w1); db1 = torch.empty_like(b1); dg = torch.empty_like(g); dbe = torch.empty_like(be); dw2 = torch.empty_like(w2); db2 = torch.empty_like(b2)
    s = _lib.stream_ptr()
    f = bench(lambda: lib.irx_mlp2_fwd(P(x), rows, din, dh, dout, P(w1), P(b1), norm, P(g), P(be), 1e-5, P(rm), P(rv), 0.1, 0.15, 1234, P(w2), P(b2), P(saved), P(y), s))
    b = bench(lambda: lib.irx_mlp2_bwd(P(x), P(dy), rows, din, dh, dout, P(w1), norm, P(g), P(w2), P(saved), 1.0 / 0.85, P(dhid), P(dx), P(dw1), P(db1), P(dg), P(dbe), P(dw2), P(db2), s))
    print("%-30s rows %3d %3d -> %3d -> %3d %s: fwd %6.1f us  bwd %6.1f us" % (name, rows, din, dh, dout, "BN" if norm == 1 else "LN", f, b))
    if not name.startswith("B="):
        tf += f; tb += b
print("sum over the seven heads: fwd %.1f us, bwd %.1f us" % (tf, tb))
