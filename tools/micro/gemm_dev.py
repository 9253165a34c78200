"""Dev micro-benchmark: device time of the small fp32 GEMMs of the language module, layouts x backends."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("USE_IRX", "1") == "1":
    import instancerefer_amd   # rocBLAS preference
dev = torch.device("cuda")
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (m, k, n) in ((480, 256, 768), (480, 300, 256), (480, 256, 256), (16, 256, 256), (64, 128, 128), (480, 768, 256)):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); wt = w.t().contiguous(); b = torch.randn(n, device=dev)
    dy = torch.randn(m, n, device=dev)
    print("m=%4d k=%4d n=%4d | linear(x,w,b) %6.1f us | addmm(b, x, w.t()) %6.1f | x@wt(NN) %6.1f | dW = dy.t()@x %6.1f | dx = dy@w %6.1f" % (
        m, k, n, bench(lambda: torch.nn.functional.linear(x, w, b)), bench(lambda: torch.addmm(b, x, w.t())), bench(lambda: x @ wt),
        bench(lambda: dy.t() @ x), bench(lambda: dy @ w)))
