"""Dev tool: ATen ops dispatched per section of a training step (forward sections, loss, backward, optimizer)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
args = bench.parse()
dev = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor
_lib.load()
import instancerefer_amd as _irx
_irx.set_compute_dtype('bf16' if args.dtype == 'bf16' else 'fp32')
B = 16
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(B, seed=123), dev)
lidar = res.pop("lidar")
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
class Counter(TorchDispatchMode):
    def __init__(self): super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.c[str(func).replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))
VIEW = ("view", "reshape", "t.", "transpose", "expand", "select", "slice", "as_strided", "unsqueeze", "squeeze", "detach", "alias", "permute", "_unsafe_view", "empty", "split", "unbind", "narrow")
def run(section, fn):
    with Counter() as cn: out = fn()
    tot = sum(cn.c.values()); views = sum(v for k, v in cn.c.items() if any(k.startswith(p) for p in VIEW))
    print("%-22s total ops %4d, non-view %4d | top: %s" % (section, tot, tot - views, ", ".join("%s:%d" % kv for kv in cn.c.most_common(9))))
    return out
for it in range(2):
    dd = dict(res); dd["irx"]._sel_cache.clear(); dd["lidar"] = SparseTensor(lidar.F, lidar.C, 1, batch_size=B)
    opt.zero_grad()
    m = model
    if it == 0:
        dd = m(dd); loss = get_loss(dd, cfg)["loss"]; loss.backward(); opt.backward_step(); continue
    dd = run("prepare", lambda: m.prepare(dd))
    dd = run("scene.encode", lambda: m.scene.encode(dd))
    dd = run("lang", lambda: m.lang(dd))
    dd = run("attribute", lambda: m.attribute(dd))
    dd = run("relation", lambda: m.relation(dd))
    dd = run("scene head", lambda: m.scene(dd))
    loss = run("get_loss", lambda: get_loss(dd, cfg)["loss"])
    run("backward", lambda: loss.backward())
    run("optimizer", lambda: opt.backward_step())
