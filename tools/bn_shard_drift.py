"""Dev: how far do the BatchNorm batch statistics of a 16-scene shard (BASELINE configs[3]: 16 scenes per GPU, per-rank
statistics) sit from those of the reference recipe's 64-scene batch (config/InstanceRefer.yaml batch_size 64)? Scene encoder
and candidate encoder, every layer: |mean_shard - mean_64| in units of the 64-batch standard deviation, and the ratio of the
standard deviations.   python tools/bn_shard_drift.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancerefer_amd import _lib, synthetic as S
from instancerefer_amd.basic_blocks import SparseConvEncoder
from instancerefer_amd.sparse import encoder_fn
from instancerefer_amd.sparse.utils import voxelize
_lib.load()
dev = torch.device("cuda")
dd = S.make_batch(64, seed=123)


def tensor_of(which, lo, hi):
    if which == "scene":
        pts = [torch.from_numpy(p) for p in dd["scene_points"][lo:hi]]
        voxel = 0.05
    else:
        pts = [torch.from_numpy(p) for ps in dd["instance_points"][lo:hi] for p in ps[:4]]
        voxel = 0.02
    allp = torch.cat(pts).to(dev)
    batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
    return voxelize(allp[:, :3].contiguous(), allp.float(), batch, [voxel] * 3, len(pts))


for which in ("scene", "candidates"):
    enc = SparseConvEncoder(7)
    enc.load_state_dict(S.seeded_state_dict(enc, 2024))
    enc = enc.to(dev).train()

    def stats(lo, hi):
        encoder_fn.TRACE = tr = {}
        with torch.enable_grad():
            enc(tensor_of(which, lo, hi))
        torch.cuda.synchronize()
        T = encoder_fn.trace_tensors(tr)
        encoder_fn.TRACE = None
        return [m.cpu().double() for m in T["mean"]], [1.0 / v.cpu().double() for v in T["invstd"]]
    mf, sf = stats(0, 64)
    dm, ds = np.zeros(13), np.zeros(13)
    for s in range(4):
        ms, ss = stats(16 * s, 16 * s + 16)
        for i in range(13):
            dm[i] = max(dm[i], float(((ms[i] - mf[i]).abs() / sf[i]).max()))
            ds[i] = max(ds[i], float((ss[i] / sf[i] - 1).abs().max()))
    print(which, "encoder: per layer, worst channel over the four 16-scene shards of a 64-scene batch")
    print("  |mean - mean64| / std64 :", " ".join("%.3f" % v for v in dm))
    print("  |std / std64 - 1|       :", " ".join("%.3f" % v for v in ds))
