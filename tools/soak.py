"""Dev tool: soak run of the pipelined training loop — memory must stay flat and the loss finite over many steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, prepare_labels
from instancerefer_amd.optim import FlatAdam
_lib.load()
import instancerefer_amd as irx
irx.set_compute_dtype('bf16' if args.dtype == 'bf16' else 'fp32')
BLOCKS = int(os.environ.get('SOAK_BLOCKS', '6'))
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
import resource
for blk in range(BLOCKS):
    t0 = time.perf_counter()
    for _ in range(100): loss = bench.step_fn(model, resident, "full", None, opt, state)
    torch.cuda.synchronize()
    print("steps %4d: %.2f ms/step, loss %.4f, cuda allocated %.1f MB, reserved %.1f MB, host RSS %.0f MB" % (
        (blk + 1) * 100, (time.perf_counter() - t0) * 10, float(loss), torch.cuda.memory_allocated() / 2**20,
        torch.cuda.memory_reserved() / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024))
