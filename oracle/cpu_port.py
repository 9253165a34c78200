"""ctypes front-end of the C/OpenMP port (oracle/csrc/spconv_cpu.c). TEST INFRASTRUCTURE / CPU baseline."""
import ctypes

import numpy as np

from . import build_c

_libs = {}


def lib(fast=False):
    """The strict-IEEE build (parity tests) or, fast=True, the -ffast-math build (bench.py's timed cpu_baseline only)."""
    if fast not in _libs:
        l = ctypes.CDLL(build_c.build(fast=fast))
        l.irx_oracle_encoder_fwd_bwd.restype = ctypes.c_double
        l.irx_oracle_encoder_fwd_bwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        l.irx_oracle_set_wgrad_double.argtypes = [ctypes.c_int]
        l.irx_oracle_encoder_param_count.restype = ctypes.c_long
        l.irx_oracle_encoder_param_count.argtypes = [ctypes.c_int]
        _libs[fast] = l
    return _libs[fast]


def pack_encoder_params(state_dict, prefix):
    """SparseConvEncoder state dict (reference key layout) -> flat float32 vector in the C port's layout."""
    order = [("stem.0.net.0", "stem.0.net.1")]
    for s in (1, 2, 3, 4):
        order += [("stage%d.0.net.0" % s, "stage%d.0.net.1" % s), ("stage%d.1.net.0" % s, "stage%d.1.net.1" % s),
                  ("stage%d.1.net.3" % s, "stage%d.1.net.4" % s)]
    parts = []
    for conv, bn in order:
        parts += [state_dict[prefix + conv + ".kernel"].detach().cpu().numpy().reshape(-1),
                  state_dict[prefix + bn + ".weight"].detach().cpu().numpy().reshape(-1),
                  state_dict[prefix + bn + ".bias"].detach().cpu().numpy().reshape(-1)]
    return np.ascontiguousarray(np.concatenate(parts).astype(np.float32)), order


def encoder_fwd_bwd(coords, feats, nbatch, params, gpool, threads=0, wgrad_double=False, fast=False):
    """coords (n,4) int32 (x,y,z,b), feats (n,c0) f32 -> (loss, pooled (nbatch,128), grads (like params)).
    wgrad_double: accumulate the weight gradients in float64 (the parity checker at full size; the timed baseline keeps
    float, torchsparse's CPU arithmetic)."""
    L = lib(fast)
    L.irx_oracle_set_wgrad_double(int(bool(wgrad_double)))
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    gpool = np.ascontiguousarray(gpool, dtype=np.float32)
    n, c0 = feats.shape
    assert params.size == L.irx_oracle_encoder_param_count(c0)
    pooled = np.zeros((nbatch, 128), np.float32)
    grads = np.zeros_like(params)
    loss = L.irx_oracle_encoder_fwd_bwd(coords.ctypes.data, feats.ctypes.data, n, c0, nbatch, params.ctypes.data,
                                            gpool.ctypes.data, pooled.ctypes.data, grads.ctypes.data, int(threads))
    return loss, pooled, grads
