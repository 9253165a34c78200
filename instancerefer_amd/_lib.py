"""ctypes binding of libirx.so (include/irx.h). Thin: argument marshalling + status -> RuntimeError.

There is deliberately NO CPU or pure-PyTorch fallback here: if the HIP library is missing or the
tensors are not on a HIP device the call raises. The CPU restatement lives in /oracle and is test
infrastructure only.
"""
import ctypes
import os

import torch

from ._build import LIB_PATH

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_Z = _c.c_size_t
_F = _c.c_float
_D = _c.c_double

# name -> (restype, argtypes) : mirrors include/irx.h one to one
_SIGNATURES = {
    "irx_version": (_I, []),
    "irx_last_error": (_c.c_char_p, []),
    "irx_device_props": (_I, [_I, _P]),
    "irx_coords_to_keys": (_I, [_P, _I, _P, _P]),
    "irx_keys_to_coords": (_I, [_P, _I, _P, _P]),
    "irx_sort_workspace_bytes": (_Z, [_I]),
    "irx_sort_pairs_u64": (_I, [_P, _I, _P, _c.c_uint64, _I, _I, _P, _P, _P, _Z, _P]),
    "irx_tile_order_workspace_bytes": (_Z, [_I]),
    "irx_tile_order": (_I, [_P, _I, _I, _I, _P, _P, _Z, _P]),
    "irx_quantize": (_I, [_P, _I, _P, _I, _D, _D, _D, _P, _P, _P]),
    "irx_hash_capacity": (_Z, [_I]),
    "irx_voxel_insert": (_I, [_P, _I, _P, _P, _Z, _P]),
    "irx_voxel_select": (_I, [_P, _I, _P, _P, _Z, _P, _P, _P]),
    "irx_hash_build": (_I, [_P, _I, _P, _P, _Z, _P]),
    "irx_kmap_build_s1": (_I, [_P, _I, _I, _P, _P, _Z, _P, _I, _P]),
    "irx_kmaps_build_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "irx_kmaps_build_pyramid": (_I, [_I, _P, _P, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P, _P]),
    "irx_downsample_workspace_bytes": (_Z, [_I]),
    "irx_downsample": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _Z, _P]),
    "irx_pyramid_build": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "irx_kmap_down_transpose": (_I, [_P, _P, _I, _P, _I, _P]),
    "irx_bev_table": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P, _I, _P, _P, _P]),
    "irx_set_compute_dtype": (_I, [_I]),
    "irx_get_compute_dtype": (_I, []),
    "irx_spconv_fwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "irx_spconv_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "irx_spconv_fwd_t": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _Z, _P]),
    "irx_spconv_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "irx_spconv_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "irx_pairs_workspace_bytes": (_Z, [_I, _I]),
    "irx_pairs_build": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _Z, _P]),
    "irx_pairs_build_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "irx_spconv_wgrad_pairs_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "irx_spconv_wgrad_pairs": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "irx_spconv_wgrad_pairs_t": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "irx_bn_workspace_bytes": (_Z, [_I, _I]),
    "irx_bn_stats": (_I, [_P, _I, _I, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "irx_bn_apply": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "irx_bn_forward": (_I, [_P, _I, _I, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "irx_bn_forward_ex": (_I, [_P, _I, _I, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _I, _I]),
    "irx_bn_backward_ex": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _I, _I, _I, _I, _I]),
    "irx_bn_backward": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "irx_bn_sums": (_I, [_P, _I, _I, _P, _P, _Z, _P]),
    "irx_bn_stats_from_sums": (_I, [_P, _D, _I, _F, _F, _P, _P, _P, _P, _P]),
    "irx_bn_backward_sums": (_I, [_P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "irx_bn_backward_apply": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _D, _P, _P, _P, _P]),
    "irx_profile_next_kernel": (_I, [_P, _P]),
    "irx_debug_set_knob": (_I, [_c.c_char_p, _c.c_long]),
    "irx_debug_get_knob": (_c.c_long, [_c.c_char_p]),
    "irx_encoder_workspace_bytes": (_Z, [_P, _P, _I, _I]),
    "irx_encoder_forward": (_I, [_P, _P, _I, _P, _Z, _P]),
    "irx_encoder_backward": (_I, [_P, _P, _I, _P, _P, _P, _Z, _P]),
    "irx_encoder_forward_sync": (_I, [_P, _P, _I, _P, _Z, _P, _P, _P, _P]),
    "irx_encoder_backward_sync": (_I, [_P, _P, _I, _P, _P, _P, _Z, _P, _P, _P, _P, _P]),
    "irx_encoder_submit": (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "irx_encoder_wait": (_I, [_I]),
    "irx_encoder_gate_next": (_I, [_I, ctypes.c_longlong, ctypes.c_ulonglong]),
    "irx_segment_max": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "irx_segment_max_backward": (_I, [_P, _P, _I, _I, _P, _P]),
    "irx_segment_mean": (_I, [_P, _I, _I, _I, _P, _P]),
    "irx_batch_offsets": (_I, [_P, _I, _I, _P, _P]),
    "irx_scene_sample": (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P, _I, _P]),
    "irx_scene_sample_batch": (_I, [_I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "irx_random_subset": (_I, [_I, _P, _I, _P, _P, _P]),
    "irx_resample_rows": (_I, [_P, _P, _I, _I, _c.c_uint64, _P, _P]),
    "irx_instance_split": (_I, [_P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P]),
    "irx_gru_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "irx_gru_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "irx_gru_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "irx_adam_step": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _P]),
    "irx_cosine_rows_fwd": (_I, [_P, _P, _P, _I, _I, _F, _P, _P, _P]),
    "irx_cosine_rows_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "irx_contrastive_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P]),
    "irx_contrastive_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "irx_total_loss": (_I, [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _I, _P, _P, _P, _P, _P]),
    "irx_attn_pool_fwd": (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "irx_attn_pool_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "irx_lang_pool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "irx_lang_pool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "irx_mlp2_saved_floats": (_Z, [_I, _I]),
    "irx_mlp2_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _F, _P, _P, _F, _F, _c.c_uint64, _P, _P, _P, _P, _P]),
    "irx_mlp2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "irx_stream_fork": (_I, [_P, _P]),
    "irx_dropout_flat": (_I, [_P, _Z, _F, _c.c_uint64, _P, _P]),
    "irx_knn_batched": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "irx_project_workspace_bytes": (_Z, [_I]),
    "irx_project_points": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "irx_project_features": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "irx_edgeconv_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "irx_edgeconv_max_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "irx_edgeconv_max_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "irx_iou_labels": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "irx_eval_select": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libirx.so (once). Raises with a build hint when it is missing — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libirx.so not found at %s — build it with `python -m instancerefer_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the irx operators." % LIB_PATH)
    # torch has already mapped its bundled libamdhip64.so (SONAME libamdhip64.so.7); the loader
    # resolves libirx's NEEDED entry against that copy, so both share one HIP runtime / streams.
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().irx_last_error()
    return msg.decode() if msg else ""


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, last_error()))


def stream_ptr() -> int:
    # raw handle of torch's current stream on the current device (fast path; no Stream object round trip)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous and on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("irx operators need tensors on a HIP device (got %s); the CPU path exists "
                           "only as the test oracle under /oracle" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("irx operators need contiguous tensors")
    return t.data_ptr()


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (name, rc, last_error()))


def set_knob(name: str, value: int):
    """Dev / test knob of the library (include/irx.h irx_debug_set_knob)."""
    call("irx_debug_set_knob", name.encode(), int(value))


def get_knob(name: str) -> int:
    return int(load().irx_debug_get_knob(name.encode()))


def hash_capacity(n: int) -> int:
    return int(load().irx_hash_capacity(int(n)))
