"""Dev tool: per-parameter gradient difference of the encoder executor between compute dtypes (dense loss)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import instancerefer_amd as irx
from helpers import device_batch, surface_cloud
from instancerefer_amd.basic_blocks import SparseConvEncoder
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(15)
clouds = [surface_cloud(rng, npts, rng.uniform(0, 3, 3), rng.uniform(0.8, 2.0, 3)) for _ in range(4)]
torch.manual_seed(2)
enc = SparseConvEncoder(7).cuda().train()
res, g = {}, None
for mode in ("fp32", "bf16_operands", "bf16"):
    irx.set_compute_dtype(mode)
    enc.zero_grad()
    out = enc(device_batch(clouds, 0.05)).F
    if g is None:
        g = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
    (out * g).sum().backward()
    torch.cuda.synchronize()
    res[mode] = (out.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters()})
irx.set_compute_dtype("fp32")
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / a.norm())
print("rows out", tuple(res["fp32"][0].shape), "out rel: op-vs-f32 %.3e st-vs-f32 %.3e st-vs-op %.3e" % (
    rel(res["fp32"][0], res["bf16_operands"][0]), rel(res["fp32"][0], res["bf16"][0]), rel(res["bf16_operands"][0], res["bf16"][0])))
for n in res["fp32"][1]:
    print("%-26s op-vs-f32 %.3e  st-vs-f32 %.3e  st-vs-op %.3e" % (n, rel(res["fp32"][1][n], res["bf16_operands"][1][n]),
          rel(res["fp32"][1][n], res["bf16"][1][n]), rel(res["bf16_operands"][1][n], res["bf16"][1][n])))
