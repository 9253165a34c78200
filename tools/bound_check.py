"""Dev tool: is the pipelined training loop host-bound or GPU-bound? After issuing step N the host checks whether the
GPU has already finished step N-1 (event.query()): if it has, the GPU ran dry waiting for the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig
from instancerefer_amd.optim import FlatAdam
_lib.load()
B = args.batch or 16
torch.manual_seed(1234)
model = bench.build_model(args, "full", dev)
bench.step_fn.cfg = DatasetConfig()
resident = S.to_device(S.make_batch(B, seed=123), dev)
lidar = resident.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F[perm].contiguous(), lidar.C[perm].contiguous(), B
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
state = {"pipeline": True, "join_wait": []}
try:
    from instancerefer_amd.loss_helper import prepare_labels
    state["labels"] = lambda dd: prepare_labels(dd, bench.step_fn.cfg, dev) if "_attr_prepared" in dd else None
except ImportError:
    pass
for _ in range(10): bench.step_fn(model, resident, "full", None, opt, state)
torch.cuda.synchronize()
N = 60
evs, dry, issue = [], 0, []
t0 = time.perf_counter()
for i in range(N):
    ts = time.perf_counter()
    bench.step_fn(model, resident, "full", None, opt, state)
    issue.append(time.perf_counter() - ts)
    e = torch.cuda.Event(); e.record(); evs.append(e)
    if i >= 1 and evs[i - 1].query(): dry += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
issue.sort()
jw = sorted(state["join_wait"][-N:])
print("join wait on the input-prep thread: median %.2f ms, p90 %.2f ms" % (jw[len(jw)//2]*1e3, jw[9*len(jw)//10]*1e3))
print("step %.2f ms | host issue per step: median %.2f ms, p10 %.2f, p90 %.2f | GPU already idle when the next step was issued: %d/%d steps"
      % (dt * 1e3, issue[N // 2] * 1e3, issue[N // 10] * 1e3, issue[9 * N // 10] * 1e3, dry, N - 1))
