"""Dev tool: in-process A/B of switches (alternating blocks of steps in ONE process on one box: the only comparison that survives the
+-3..5 % box-to-box / run-to-run spread of this loop). Each switch is  module:ATTR  (a module-level boolean, e.g.
instancerefer_amd.scene_module:_FUSED_ATTN)  or  knob:NAME  (irx_debug_set_knob, e.g. knob:enc_fold_slabs).
usage: python tools/ab_inproc.py [--dtype bf16|f32] [--reps 6] [--block 40] SWITCH [SWITCH ...]     (each switch A/B'd in turn)"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--block", type=int, default=40)
ap.add_argument("switches", nargs="+")
a = ap.parse_args()
sys.argv = ["bench.py"]
import torch, bench
import instancerefer_amd as irx
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, prepare_labels
from instancerefer_amd.optim import FlatAdam
_lib.load()
irx.set_compute_dtype("bf16" if a.dtype == "bf16" else "fp32")
B = 16
torch.manual_seed(1234)
model = bench.build_model(args, "full", dev)
bench.step_fn.cfg = DatasetConfig()
resident = S.to_device(S.make_batch(B, seed=123), dev)
lidar = resident.pop("lidar"); perm = torch.randperm(lidar.F.shape[0], device=dev)
resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F[perm].contiguous(), lidar.C[perm].contiguous(), B
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, module=model)
state = {"pipeline": True, "threaded": False, "at_backward": True}
state["labels"] = lambda dd: prepare_labels(dd, bench.step_fn.cfg, dev) if "_attr_prepared" in dd else None


def setter(sw):
    kind, name = sw.split(":", 1)
    if kind == "knob":
        val = 1
        if "=" in name:
            name, val = name.split("=")[0], int(name.split("=")[1])
        return lambda on: _lib.set_knob(name, val if on else 0)
    mod = importlib.import_module(kind)
    if "=" in name:                                   # module:ATTR=a|b  -> integer a when off, b when on
        name, vals = name.split("=")
        a_, b_ = (int(v) for v in vals.split("|"))
        return lambda on: setattr(mod, name, b_ if on else a_)
    return lambda on: setattr(mod, name, bool(on))


def block(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bench.step_fn(model, resident, "full", None, opt, state)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


block(40)
for sw in a.switches:
    set_ = setter(sw)
    for flag in (False, True):
        set_(flag); block(12)
    res = {False: [], True: []}
    for rep in range(a.reps):
        for flag in ((False, True) if rep % 2 == 0 else (True, False)):
            set_(flag); block(3)
            res[flag].append(block(a.block))
    med = {}
    for flag in (False, True):
        v = sorted(res[flag]); med[flag] = v[len(v) // 2]
        print("%s=%-5s ms/step: %s | median %.3f" % (sw, flag, " ".join("%.2f" % x for x in res[flag]), med[flag]))
    print("  -> %s on/off: %+.2f %% step time" % (sw, 100.0 * (med[True] / med[False] - 1.0)))
    set_(True)
