"""Dev: which objects of a training step are only freed by the CYCLIC collector? 20 steps with automatic collection off, then one
collection with DEBUG_SAVEALL: the types (and, for containers, a hint of their content) of what it found."""
import argparse, collections, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0)
import instancerefer_amd as irx
from instancerefer_amd import synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, prepare_labels
from instancerefer_amd.optim import FlatAdam
irx.set_compute_dtype("bf16")
model = bench.build_model(args, "full", dev)
bench.step_fn.cfg = DatasetConfig()
res = S.to_device(S.make_batch(16, seed=123), dev)
lidar = res.pop("lidar")
res["lidar_F"], res["lidar_C"], res["B"] = lidar.F, lidar.C, 16
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, module=model)
state = {"pipeline": True, "threaded": False, "at_backward": True}
state["labels"] = lambda dd: prepare_labels(dd, bench.step_fn.cfg, dev) if "_attr_prepared" in dd else None
for _ in range(10):
    bench.step_fn(model, res, "full", None, opt, state)
torch.cuda.synchronize()
gc.collect(); gc.freeze(); gc.disable()
N = 20
for _ in range(N):
    bench.step_fn(model, res, "full", None, opt, state)
torch.cuda.synchronize()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print("unreachable objects after %d steps: %d (%.0f per step)" % (N, n, n / N))
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
for k, v in cnt.most_common(25):
    print("  %-40s %6d  %.1f / step" % (k, v, v / N))
# hints: dict keys / function names / cell contents
hint = collections.Counter()
for o in gc.garbage:
    if isinstance(o, dict):
        hint["dict keys: " + ",".join(sorted(str(k) for k in list(o.keys())[:6]))[:110]] += 1
    elif type(o).__name__ == "function":
        hint["function: " + o.__qualname__] += 1
    elif type(o).__name__ == "cell":
        try:
            hint["cell -> " + type(o.cell_contents).__name__] += 1
        except ValueError:
            pass
for k, v in hint.most_common(40):
    print("  %5d  %s" % (v, k))
