"""Dev tool: where the host is at each moment of the REAL step (no syncs added): entry / exit clocks of the model's phases,
the lane waits and the encoders' backward submissions, relative to the step start; plus GPU-side timestamps of the
same marks (events on the current stream) to see when the GPU gets there.  python tools/phase_clock.py [--dtype bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
import instancerefer_amd as irx
from instancerefer_amd import loss_helper
from instancerefer_amd.loss_helper import DatasetConfig
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor, encoder_fn
_lib.load()
irx.set_compute_dtype({"f32": "fp32", "bf16": "bf16", "bf16op": "bf16_operands"}[args.dtype])
B = args.batch or 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
F_, C_ = lidar.F[perm].contiguous(), lidar.C[perm].contiguous()
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, module=model)
marks = []
T0 = [0.0]
def mark(name):
    marks.append((name, (time.perf_counter() - T0[0]) * 1e3))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        mark(name + " >")
        r = f(*a, **k)
        mark(name + " <")
        return r
    setattr(obj, attr, g)
wrap(model.scene, "encode", "scene.encode")
if hasattr(model.attribute, "encode"): wrap(model.attribute, "encode", "attr.encode")
wrap(model.lang, "forward", "lang")
wrap(model.attribute, "forward", "attribute")
wrap(model.relation, "forward", "relation")
wrap(model.scene, "forward", "scene head")
wrap(model, "prepare", "prepare")
_lw = encoder_fn.lane_wait
def lw(lane):
    mark("lane_wait >"); _lw(lane); mark("lane_wait <")
encoder_fn.lane_wait = lw
import instancerefer_amd.scene_module as sm, instancerefer_amd.instancerefer as im, instancerefer_amd.attribute_module as am
for m in (sm, im, am):
    if hasattr(m, "lane_wait"): m.lane_wait = lw
_bw = encoder_fn.EncoderFn.backward
def bw(ctx, dout):
    mark("enc.backward >"); r = _bw(ctx, dout); mark("enc.backward <"); return r
encoder_fn.EncoderFn.backward = staticmethod(bw)
acc = {}
for it in range(12):
    dd = dict(res); dd["irx"]._sel_cache.clear()
    dd["lidar"] = SparseTensor(F_, C_, 1, batch_size=B)
    torch.cuda.synchronize()
    marks.clear(); T0[0] = time.perf_counter()
    opt.zero_grad(); dd = model(dd); mark("forward done")
    loss = loss_helper.get_loss(dd, cfg)["loss"]; mark("loss done")
    loss.backward(); mark("backward returned")
    opt.backward_step(); mark("optimizer issued")
    torch.cuda.synchronize(); mark("gpu drained")
    if it >= 4:
        seen = {}
        for n, t in marks:
            k = seen.get(n, 0); seen[n] = k + 1
            acc.setdefault((n, k), []).append(t)
order = sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1]))
for (n, k), v in order:
    print("%8.2f ms  %s%s" % (sum(v) / len(v), n, "" if k == 0 else " #%d" % (k + 1)))
