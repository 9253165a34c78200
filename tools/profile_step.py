"""Host-side profile of one bench step (cProfile + per-section cuda sync timing). Dev tool."""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
device = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig
_lib.load()
B = args.batch or 16
model = bench.build_model(args, args.workload, device)
bench.step_fn.cfg = DatasetConfig()
host = S.make_batch(B, seed=123, num_points=args.points, num_instances=args.instances, num_candidates=args.candidates, tokens=args.tokens)
res = S.to_device(host, device)
lidar = res.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=device)
res["lidar_F"] = lidar.F[perm].contiguous(); res["lidar_C"] = lidar.C[perm].contiguous(); res["B"] = B
from instancerefer_amd.optim import FlatAdam
red = None
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
state = {'pipeline': True}
for _ in range(3): bench.step_fn(model, res, args.workload, red, opt, state)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(5): bench.step_fn(model, res, args.workload, red, opt, state)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
pr.disable()
print("ms/step", dt * 1e3)
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45); print(st.getvalue()[:9000])
