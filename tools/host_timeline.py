"""Dev tool: host-side issue time of each phase WITHOUT syncs, plus the final GPU drain time."""
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
acc = {}
for it in range(10):
    dd = dict(res); dd["irx"]._sel_cache.clear()
    dd["lidar"] = SparseTensor(F_, C_, 1, batch_size=B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(); dd = model(dd); t1 = time.perf_counter()
    loss = get_loss(dd, cfg)["loss"]; t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    opt.backward_step(); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    if it >= 3:
        for k, v in (("host fwd issue", t1 - t0), ("host loss issue", t2 - t1), ("host bwd issue", t3 - t2), ("host opt issue", t4 - t3), ("gpu drain after host done", t5 - t4), ("total", t5 - t0)):
            acc.setdefault(k, []).append(v * 1e3)
for k, v in acc.items(): print("%-28s %7.2f ms" % (k, sum(v) / len(v)))
