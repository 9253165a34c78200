"""Dev tool: cProfile of the host side of a training step (no syncs inside the profiled region)."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor
_lib.load()
import instancerefer_amd as _irx
_irx.set_compute_dtype('bf16' if args.dtype == 'bf16' else 'fp32')
B = args.batch or 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
F_, C_ = lidar.F[perm].contiguous(), lidar.C[perm].contiguous()
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
pr = cProfile.Profile()
N = 8
for it in range(4 + N):
    dd = dict(res); dd["irx"]._sel_cache.clear()
    dd["lidar"] = SparseTensor(F_, C_, 1, batch_size=B)
    torch.cuda.synchronize()
    if it >= 4: pr.enable()
    opt.zero_grad(); dd = model(dd)
    loss = get_loss(dd, cfg)["loss"]
    loss.backward()
    opt.backward_step()
    if it >= 4: pr.disable()
    torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    txt = s.getvalue()
    print("\n".join(l[:170] for l in txt.splitlines()[:75]))
print("per-step divide by", N)
