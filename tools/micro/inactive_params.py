"""Dev: which parameters of the bench model never receive a gradient (FlatAdam._inactive), and what the forced one-rank
collective path costs per phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py"]
import torch, bench, torch.distributed as dist
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor
_lib.load()
model = bench.build_model(args, "full", dev)
res = S.to_device(S.make_batch(16, seed=123), dev)
lidar = res.pop("lidar")
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, module=model)
opt._force_collectives = True
cfg = DatasetConfig()
names = {id(p): n for n, p in model.named_parameters()}
for it in range(6):
    dd = dict(res); dd["irx"]._sel_cache.clear(); dd["lidar"] = SparseTensor(lidar.F, lidar.C, 1, batch_size=16)
    opt.zero_grad(); loss = get_loss(model(dd), cfg)["loss"]; loss.backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.gather_grads(); t1 = time.perf_counter()
    inactive = sorted(opt._inactive)
    opt.all_reduce(); t2 = time.perf_counter()
    opt.step(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("step %d: gather %.2f ms, all_reduce %.2f ms, adam issue %.2f ms, drain %.2f ms; inactive %d: %s" % (
        it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, len(inactive), [names[id(opt.params[i])] for i in inactive][:8]))
print("groups", [(a, b) for _, a, b in opt._groups], "gaps", opt._gaps)
dist.barrier(); dist.destroy_process_group()
