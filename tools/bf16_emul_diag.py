"""Dev: layer-by-layer forward comparison of the product encoder (per-layer path) with the emulating oracle in a bf16 mode.
   python tools/bf16_emul_diag.py [bf16_operands|bf16] [c0]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import align, device_batch, oracle_batch, surface_cloud  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16_operands"
c0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
import instancerefer_amd as irx  # noqa: E402
from instancerefer_amd import synthetic as S  # noqa: E402
from instancerefer_amd.basic_blocks import SparseConvEncoder  # noqa: E402
from instancerefer_amd.sparse import encoder_fn, nn as spnn  # noqa: E402
from instancerefer_amd.sparse.tensor import SparseTensor  # noqa: E402
from oracle.model_ref import SparseConvEncoder as OE  # noqa: E402
from oracle.torchsparse import SparseTensor as OT  # noqa: E402
from oracle.torchsparse.nn import emulate  # noqa: E402

rng = np.random.default_rng(15 + c0)
clouds = [surface_cloud(rng, 4000, rng.uniform(0, 3, 3), rng.uniform(0.8, 2.0, 3), c_extra=c0 - 3) for _ in range(4)]
enc = SparseConvEncoder(c0)
sd = S.seeded_state_dict(enc, 31 + c0)
enc.load_state_dict(sd)
ora = OE(c0)
ora.load_state_dict(sd)
enc = enc.cuda().train()
ora.train()
d = device_batch(clouds, 0.05)
o = oracle_batch(clouds, 0.05)


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


irx.set_compute_dtype(mode)
xd, xo = d, o
outs_d, outs_o = [], []
with emulate.mode(mode), torch.no_grad():
    sk_d, sk_o = encoder_fn._skeleton(enc), encoder_fn._skeleton(ora)
    for i, ((cd, bd, down, res), (co, bo, _, _)) in enumerate(zip(sk_d, sk_o)):
        yd, lv = cd.conv_feats(xd)
        yo = co(xo)
        ia, ib = align(lv.coords.cpu().numpy(), yo.C.numpy())
        e_conv = rel(yd.cpu()[ia], yo.F[ib])
        # the same conv inputs through the fp32 oracle conv, for scale
        with emulate.mode(None):
            y32 = co(xo)
        e_32 = rel(yd.cpu()[ia], y32.F[ib])
        rd = outs_d[res].F if res >= 0 else None
        zd = bd.feats(yd, rd, True)
        zo = bo(yo)
        if res >= 0:
            zo = zo + outs_o[res]
        zo = OT(torch.relu(zo.F), zo.C, zo.s)
        zo.kernel_maps, zo.coord_maps = yo.kernel_maps, yo.coord_maps
        e_out = rel(zd.cpu()[ia], zo.F[ib])
        print("layer %2d %3d->%3d K=%2d n=%6d  conv vs emu %.2e (vs fp32 oracle %.2e)  after bn/relu %.2e" % (
            i, cd.kernel.shape[1], cd.kernel.shape[2], cd.kernel.shape[0], lv.n, e_conv, e_32, e_out))
        xd = SparseTensor(zd, lv.coords, lv.stride, lv.batch_size, lv)
        xo = zo
        outs_d.append(xd)
        outs_o.append(xo)
irx.set_compute_dtype("fp32")
