"""Dev: cProfile of EncoderFn.forward / backward (the Python between autograd and the library's encoder executor)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py"] + sys.argv[1:]
import torch, bench
args = bench.parse()
dev = torch.device("cuda", 0)
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam
from instancerefer_amd.sparse import SparseTensor, encoder_fn
import instancerefer_amd as irx
_lib.load()
irx.set_compute_dtype('bf16' if args.dtype == 'bf16' else 'fp32')
model = bench.build_model(args, "full", dev)
cfg = DatasetConfig()
res = S.to_device(S.make_batch(16, seed=123), dev)
lidar = res.pop("lidar")
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, module=model)
prf, prb = cProfile.Profile(), cProfile.Profile()
on = [False]
of, ob = encoder_fn.EncoderFn.forward, encoder_fn.EncoderFn.backward
def f(ctx, *a):
    if on[0]: prf.enable()
    r = of(ctx, *a)
    if on[0]: prf.disable()
    return r
def b(ctx, *a):
    if on[0]: prb.enable()
    r = ob(ctx, *a)
    if on[0]: prb.disable()
    return r
encoder_fn.EncoderFn.forward = staticmethod(f); encoder_fn.EncoderFn.backward = staticmethod(b)
N = 10
for it in range(4 + N):
    on[0] = it >= 4
    dd = dict(res); dd["irx"]._sel_cache.clear(); dd["lidar"] = SparseTensor(lidar.F, lidar.C, 1, batch_size=16)
    opt.zero_grad(); loss = get_loss(model(dd), cfg)["loss"]; loss.backward(); opt.backward_step()
    torch.cuda.synchronize()
for name, pr in (("forward", prf), ("backward", prb)):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("==== EncoderFn.%s (2 calls per step, %d steps)" % (name, N)); print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
