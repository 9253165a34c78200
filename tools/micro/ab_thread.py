"""Dev tool: in-process A/B of a module-level boolean switch (alternating blocks of steps in ONE process on one box:
the only comparison that survives the +-5 % box-to-box / run-to-run spread of the host-bound loop).
usage: python tools/ab_inproc.py instancerefer_amd.sparse.tensor PREBUILD_TABLES"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
modname, attr = "bench", "threaded-pipeline"
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, prepare_labels
from instancerefer_amd.optim import FlatAdam
_lib.load()
B = 16
torch.manual_seed(1234)
model = bench.build_model(args, "full", dev)
bench.step_fn.cfg = DatasetConfig()
resident = S.to_device(S.make_batch(B, seed=123), dev)
lidar = resident.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F[perm].contiguous(), lidar.C[perm].contiguous(), B
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
state = {"pipeline": True, "threaded": True}
state["labels"] = lambda dd: prepare_labels(dd, bench.step_fn.cfg, dev) if "_attr_prepared" in dd else None
def block(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bench.step_fn(model, resident, "full", None, opt, state)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for flag in (False, True):
    state["threaded"] = flag; block(12)
res = {False: [], True: []}
for rep in range(5):
    for flag in (False, True):
        state["threaded"] = flag; block(3)
        res[flag].append(block(40))
for flag in (False, True):
    v = sorted(res[flag])
    print("%s=%-5s ms/step: %s | median %.2f" % (attr, flag, " ".join("%.2f" % x for x in res[flag]), v[len(v) // 2]))
