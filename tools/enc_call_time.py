"""Dev tool: host time spent INSIDE the irx_encoder_forward / irx_encoder_backward C calls per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor
lib = _lib.load()
acc = {}
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); acc.setdefault(self.name, []).append(time.perf_counter() - t); return r
for n in ("irx_encoder_forward", "irx_encoder_backward", "irx_encoder_workspace_bytes", "irx_pairs_build_multi"):
    object.__setattr__(lib, n, Timed(n, getattr(lib, n)))
B = 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
F_, C_ = lidar.F[perm].contiguous(), lidar.C[perm].contiguous()
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
N = 10
for it in range(4 + N):
    if it == 4: acc.clear()
    dd = dict(res); dd["irx"]._sel_cache.clear()
    dd["lidar"] = SparseTensor(F_, C_, 1, batch_size=B)
    opt.zero_grad(); dd = model(dd)
    loss = get_loss(dd, cfg)["loss"]; loss.backward(); opt.backward_step()
    torch.cuda.synchronize()
for k, v in acc.items():
    print("%-30s %5.1f calls/step  %7.1f us each  %6.3f ms/step" % (k, len(v) / N, 1e6 * sum(v) / len(v), 1e3 * sum(v) / N))
