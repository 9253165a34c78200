"""Dev tool: torch.profiler CPU-side op table of a training step (which ops cost host time, fwd and bwd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse()
dev = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor
_lib.load()
B = args.batch or 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
F_, C_ = lidar.F[perm].contiguous(), lidar.C[perm].contiguous()
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
def step():
    dd = dict(res); dd["irx"]._sel_cache.clear()
    dd["lidar"] = SparseTensor(F_, C_, 1, batch_size=B)
    opt.zero_grad(); dd = model(dd)
    loss = get_loss(dd, cfg)["loss"]
    loss.backward()
    opt.backward_step()
for _ in range(4): step()
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=60))
print("steps:", N)
