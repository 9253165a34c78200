"""Loader of csrc/_irx_nodes.so — the C++ autograd nodes over libirx's C-ABI (csrc/torch_nodes.cpp).

The module is handed the ADDRESSES of the entry points of the library instance _lib.py loaded (it links nothing itself), so
a dev build selected with IRX_LIB_PATH is the one the nodes call. IRX_CPP_NODES=0 keeps every head on its Python / ATen path.
"""
import ctypes
import importlib.util
import os

from . import _build, _lib

ENABLED = os.environ.get("IRX_CPP_NODES", "1") != "0"
_ENTRY_POINTS = ("irx_mlp2_saved_floats", "irx_mlp2_fwd", "irx_mlp2_bwd", "irx_gru_forward", "irx_gru_backward", "irx_gru_wgrad", "irx_last_error",
                 "irx_hash_capacity", "irx_hash_build", "irx_kmap_build_s1", "irx_kmaps_build_multi", "irx_kmaps_build_pyramid",
                 "irx_pairs_workspace_bytes", "irx_pairs_build_multi", "irx_kmap_down_transpose")
# csrc/heads_nodes.cpp (one node per head)
_HEAD_ENTRY_POINTS = ("irx_mlp2_saved_floats", "irx_mlp2_fwd", "irx_mlp2_bwd", "irx_segment_max", "irx_segment_max_backward",
                      "irx_cosine_rows_fwd", "irx_cosine_rows_bwd", "irx_spconv_fwd_workspace_bytes", "irx_spconv_fwd",
                      "irx_spconv_wgrad_workspace_bytes", "irx_spconv_wgrad", "irx_bn_workspace_bytes", "irx_bn_forward", "irx_bn_backward",
                      "irx_kmap_down_transpose", "irx_attn_pool_fwd", "irx_attn_pool_bwd", "irx_dropout_flat", "irx_stream_fork", "irx_total_loss", "irx_edgeconv_workspace_bytes",
                      "irx_edgeconv_max_fwd", "irx_edgeconv_max_bwd", "irx_lang_pool_fwd", "irx_lang_pool_bwd")
_mod = None
_tried = False
_lock = __import__("threading").Lock()


def load():
    """The extension module, bound to libirx — or None when it is switched off or has not been built (the callers then use
    their Python autograd.Function: same kernels, more interpreter time; nothing here is a CPU fallback). Thread-safe (the language
    module's helper thread reaches it too); a module OLDER than csrc/torch_nodes.cpp is refused with a warning — it would call the
    C-ABI through the function types of another revision."""
    global _mod, _tried
    if _tried:
        return _mod
    with _lock:
        if _tried:
            return _mod
        mod = _load_locked()
        _mod, _tried = mod, True
        return mod


def _load_locked():
    if not ENABLED or not os.path.exists(_build.NODES_PATH):
        return None
    if _build.nodes_stale():
        import warnings
        warnings.warn("instancerefer_amd: csrc/_irx_nodes.so is older than its sources (csrc/torch_nodes.cpp, heads_nodes.cpp) — the dense heads run through their "
                      "Python / ATen path; rebuild with `python -m instancerefer_amd._build`", RuntimeWarning)
        return None
    try:
        spec = importlib.util.spec_from_file_location("_irx_nodes", _build.NODES_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        lib = _lib.load()
        mod.bind({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in _ENTRY_POINTS})
        mod.bind_heads({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in _HEAD_ENTRY_POINTS})
    except Exception as e:                 # a module built against another torch: say so, run the heads through their Python nodes
        import warnings
        warnings.warn("instancerefer_amd: csrc/_irx_nodes.so could not be loaded (%s: %s) — the dense heads run through their "
                      "Python / ATen path; rebuild with `python -m instancerefer_amd._build`" % (type(e).__name__, e), RuntimeWarning)
        return None
    return mod
