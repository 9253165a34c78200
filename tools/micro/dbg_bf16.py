import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import instancerefer_amd as irx
from instancerefer_amd import _lib
from instancerefer_amd.sparse import functional as F_
from helpers import surface_cloud, device_batch
_lib.load()
rng = np.random.default_rng(0)
clouds = [surface_cloud(rng, 3000)]
d = device_batch(clouds, 0.05)
lv = d.level(); tbl, ld = lv.nbr27(); n = lv.n
for cin, cout in ((32, 32), (128, 128)):
    x = torch.ones(n, cin, device="cuda")
    w = torch.zeros(27, cin, cout, device="cuda"); 
    for k in range(27): w[k] = 0.01 * (k + 1)
    w[:, 1, :] += 0.5            # asymmetric in c
    w[:, :, 3] += 0.25           # asymmetric in n
    y32 = F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 0, 0)
    irx.set_compute_dtype("bf16")
    yb = F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 0, 0)
    irx.set_compute_dtype("fp32")
    print(cin, cout, "fp32", y32[0, :6].tolist(), "\n   bf16", yb[0, :6].tolist(), "\n   maxdiff", (y32 - yb).abs().max().item(), "finite", torch.isfinite(yb).all().item())
