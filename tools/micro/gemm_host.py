import time, torch
dev = torch.device('cuda')
a = torch.randn(64, 256, device=dev); b = torch.randn(256, 128, device=dev); bias = torch.randn(128, device=dev)
lin = torch.nn.Linear(256, 128).to(dev)
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, fn in (("mm", lambda: torch.mm(a, b)), ("addmm", lambda: torch.addmm(bias, a, b)), ("linear", lambda: lin(a))):
        for _ in range(50): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-9s %-7s host %.2f us/call  (drain %.2f us/call)" % (lib, name, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
x = torch.randn(64, 128, device=dev)
for name, fn in (("relu", lambda: torch.relu(x)), ("add", lambda: x + x), ("empty", lambda: torch.empty(64, 128, device=dev))):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-9s host %.2f us/call (drain %.2f)" % (name, (t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
