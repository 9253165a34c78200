"""Dev tool: host time spent inside the Python backward of each custom autograd Function (they run on the autograd
engine's thread, invisible to cProfile of the main thread) vs the whole run_backward call."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from instancerefer_amd import _lib, synthetic as S, dense
from instancerefer_amd.sparse import functional as F_, encoder_fn
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor
_lib.load()
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, which):
    orig = getattr(cls, which)
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); dt = time.perf_counter() - t0
        e = acc[cls.__name__ + "." + which]; e[0] += dt; e[1] += 1
        return r
    setattr(cls, which, staticmethod(timed))
classes = [encoder_fn.EncoderFn, dense.GRULayerFn] + [v for v in vars(F_).values() if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function]
for c in classes:
    wrap(c, "backward"); wrap(c, "forward")
B = 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar")
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
tb = []
N = 10
for it in range(4 + N):
    dd = dict(res); dd["irx"]._sel_cache.clear(); dd["lidar"] = SparseTensor(lidar.F, lidar.C, 1, batch_size=B)
    if it == 4:
        acc.clear(); tb = []
    opt.zero_grad(); dd = model(dd); loss = get_loss(dd, cfg)["loss"]
    t0 = time.perf_counter(); loss.backward(); tb.append(time.perf_counter() - t0)
    opt.backward_step(); torch.cuda.synchronize()
print("run_backward: %.2f ms/step" % (sum(tb) / N * 1e3))
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-32s %6.3f ms/step (%4.1f calls/step, %6.1f us each)" % (k, t / N * 1e3, n / N, t / n * 1e6))
