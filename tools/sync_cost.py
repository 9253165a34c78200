"""Dev tool: host time blocked in Tensor.item() (the level-size / voxel-count syncs of prepare()) per step of the
pipelined loop, and the total time of prepare()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
state = {"pipeline": True, "threaded": False}
state["labels"] = lambda dd: prepare_labels(dd, bench.step_fn.cfg, dev) if "_attr_prepared" in dd else None
acc = {"item": 0.0, "n": 0, "prep": 0.0}
orig_item = torch.Tensor.item
def timed_item(self):
    t0 = time.perf_counter(); r = orig_item(self); acc["item"] += time.perf_counter() - t0; acc["n"] += 1; return r
orig_l, orig_f = model.prepare_launch, model.prepare_finish
acc["fin"] = 0.0
def timed_l(dd):
    t0 = time.perf_counter(); r = orig_l(dd); acc["prep"] += time.perf_counter() - t0; return r
def timed_f(dd):
    t0 = time.perf_counter(); r = orig_f(dd); dt = time.perf_counter() - t0; acc["prep"] += dt; acc["fin"] += dt; return r
for _ in range(10): bench.step_fn(model, resident, "full", None, opt, state)
torch.cuda.synchronize()
torch.Tensor.item = timed_item; model.prepare_launch = timed_l; model.prepare_finish = timed_f
N = 50
t0 = time.perf_counter()
for _ in range(N): bench.step_fn(model, resident, "full", None, opt, state)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("step %.2f ms | prepare launch+finish %.2f ms/step (finish phase %.2f ms), of which blocked in .item(): %.2f ms (%.1f syncs/step, %.0f us each)"
      % (dt * 1e3, acc["prep"] / N * 1e3, acc["fin"] / N * 1e3, acc["item"] / N * 1e3, acc["n"] / N, acc["item"] / max(acc["n"], 1) * 1e6))
