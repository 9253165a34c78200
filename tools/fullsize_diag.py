"""Dev tool: per-parameter gradient error of the HIP scene encoder vs the C/OpenMP port at 16 x 50 k points, with float and
double weight-gradient accumulation in the port, plus HIP run-to-run determinism."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_port
from instancerefer_amd import synthetic as S
from instancerefer_amd.basic_blocks import SparseConvEncoder
from instancerefer_amd.sparse import nn as spnn
from instancerefer_amd.sparse.utils import voxelize
dev = torch.device("cuda")
dd = S.make_batch(16, seed=321)
pts = [torch.from_numpy(p) for p in dd["scene_points"]]
allp = torch.cat(pts).to(dev)
batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 16)
enc = SparseConvEncoder(7); enc.load_state_dict(S.seeded_state_dict(enc, 4242)); enc = enc.to(dev).train()
g = torch.from_numpy(np.random.default_rng(5).standard_normal((16, 128)).astype(np.float32))
runs = []
for r in range(2):
    enc.zero_grad()
    pooled = spnn.GlobalMaxPooling()(enc(st))
    (pooled * g.to(dev)).sum().backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().cpu().numpy().reshape(-1).copy() for n, p in enc.named_parameters()})
print("HIP deterministic:", all(np.array_equal(runs[0][n], runs[1][n]) for n in runs[0]))
params, order = cpu_port.pack_encoder_params({k: v.detach().cpu() for k, v in enc.state_dict().items()}, "")
for dbl in (False, True):
    closs, cpooled, cgrads = cpu_port.encoder_fwd_bwd(st.C.cpu().numpy(), st.F.detach().cpu().numpy(), 16, params, g.numpy(), wgrad_double=dbl)
    print("wgrad_double", dbl, "pooled err", float(np.abs(pooled.detach().cpu().numpy() - cpooled).max()))
    off = 0
    for conv, bn in order:
        for name in (conv + ".kernel", bn + ".weight", bn + ".bias"):
            ref = cgrads[off:off + runs[0][name].size]; off += ref.size
            got = runs[0][name]
            print("  %-24s max|ref| %10.4f  maxerr/max %9.2e  l2rel %9.2e" % (name, np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max(), np.linalg.norm(got - ref) / np.linalg.norm(ref)))
