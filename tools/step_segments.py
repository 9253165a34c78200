"""Dev tool: wall time of the step's segments with a device sync after each (host+GPU serialized per segment),
and GPU-only time of each segment via events (async issue)."""
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
B = args.batch or 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
F_, C_ = lidar.F[perm].contiguous(), lidar.C[perm].contiguous()
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
def sync(): torch.cuda.synchronize()
seg = {}
def T(name, t0):
    sync(); seg.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
for it in range(8):
    dd = dict(res); dd["irx"]._sel_cache.clear()
    dd["lidar"] = SparseTensor(F_, C_, 1, batch_size=B)
    opt.zero_grad(); sync()
    t = time.perf_counter(); dd = model.lang(dd); T("lang", t)
    t = time.perf_counter(); dd = model.attribute(dd); T("attribute", t)
    t = time.perf_counter(); dd = model.relation(dd); T("relation", t)
    t = time.perf_counter(); dd = model.scene(dd); T("scene", t)
    t = time.perf_counter(); loss = get_loss(dd, cfg)["loss"]; T("loss", t)
    t = time.perf_counter(); loss.backward(); T("backward", t)
    t = time.perf_counter(); opt.backward_step(); T("optimizer", t)
tot = 0
for k, v in seg.items():
    m = sum(v[3:]) / len(v[3:]); tot += m
    print("%-10s %7.2f ms" % (k, m))
print("sum        %7.2f ms" % tot)
