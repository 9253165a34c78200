"""The C-ABI library loads on a GPU-less host and exports every symbol include/irx.h declares; pure-host entry
points (version, capacities, workspace sizes, argument validation + error strings) behave. No compute calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "irx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(irx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    from instancerefer_amd import _lib
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libirx.so does not export %s" % n
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, set(names) ^ set(_lib.EXPORTED_SYMBOLS)


def test_encoder_descriptor_fields_match_header():
    """The Python executor fills the descriptor table by field index: the order must be the header's enum."""
    from instancerefer_amd.sparse.encoder_fn import ENC_FIELDS
    src = open(os.path.join(ROOT, "include", "irx.h")).read()
    body = re.search(r"enum\s*\{(.*?IRX_ENC_NFIELDS)\s*\}", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n for n in re.findall(r"IRX_ENC_([A-Z0-9_]+)", body) if n != "NFIELDS"]
    assert tuple(names) == ENC_FIELDS


def test_encoder_executor_validates(lib):
    assert lib.irx_encoder_workspace_bytes(None, None, 0, 0) == 0
    assert lib.irx_encoder_forward(None, None, 0, None, 0, None) == -1
    assert b"irx_encoder_forward" in lib.irx_last_error()


def test_host_only_entry_points(lib):
    assert lib.irx_version() == 1
    assert lib.irx_hash_capacity(1000) == 2048 and lib.irx_hash_capacity(0) == 64
    assert lib.irx_hash_capacity(1 << 20) == 1 << 21
    assert lib.irx_bn_workspace_bytes(1000, 128) == 16 * 2 * 128 * 4      # 64 rows per statistics workgroup (256 until round 5)
    assert lib.irx_downsample_workspace_bytes(5000) >= 3 * 4
    small = lib.irx_spconv_wgrad_workspace_bytes(100, 27, 128, 128)
    assert lib.irx_spconv_wgrad_workspace_bytes(5000, 27, 7, 32) == 20 * 27 * 7 * 32 * 4   # stem path: per-workgroup partials
    big = lib.irx_spconv_wgrad_workspace_bytes(500000, 27, 128, 128)
    assert small == 0 and big > 0 and big % (27 * 128 * 128 * 4) == 0


def test_argument_validation_and_error_string(lib):
    rc = lib.irx_spconv_fwd(None, None, None, 0, 10, 27, 0, 32, 0, 0, None, None, 0, None)
    assert rc == -1
    assert b"irx_spconv_fwd" in lib.irx_last_error()
    rc = lib.irx_kmap_build_s1(None, 5, 3, None, None, 64, None, 5, None)   # stride 3 is not a power of two
    assert rc == -1 and b"power of two" in lib.irx_last_error()
    assert lib.irx_spconv_fwd(None, None, None, 0, 0, 27, 7, 32, 0, 0, None, None, 0, None) == 0   # empty input is fine
    out = (ctypes.c_int * 8)()
    assert lib.irx_device_props(0, out) in (0, -4)


def test_mode_switch_and_lane_calls_validate_their_arguments(lib):
    assert lib.irx_get_compute_dtype() == 0                      # fp32 (exact) is the default
    assert lib.irx_set_compute_dtype(3) == -1 and b"irx_set_compute_dtype" in lib.irx_last_error()
    assert lib.irx_set_compute_dtype(2) == 0 and lib.irx_get_compute_dtype() == 2     # bf16 storage in the executor
    assert lib.irx_set_compute_dtype(0) == 0
    assert lib.irx_set_compute_dtype(1) == 0 and lib.irx_get_compute_dtype() == 1
    assert lib.irx_set_compute_dtype(0) == 0 and lib.irx_get_compute_dtype() == 0
    assert lib.irx_encoder_wait(0) == 0                          # an idle lane returns at once
    assert lib.irx_encoder_wait(7) == -1 and b"lane" in lib.irx_last_error()
    assert lib.irx_encoder_submit(0, 0, None, None, 0, None, None, None, 0, None) == -1
    assert lib.irx_scene_sample(None, 0, 7, None, 0, 0, 0, None, 0, None, None, 4, None) == 0    # empty sample
    assert lib.irx_scene_sample(None, 0, 7, None, 0, 0, 0, None, 0, None, None, 2, None) == -1   # elem_bytes
    assert lib.irx_instance_split(None, 0, 7, None, None, 0, None, 1024, None, None, None, 8, None) == 0


def test_product_never_imports_oracle():
    """The product path must not route through the oracle or any CPU fallback."""
    pkg = os.path.join(ROOT, "instancerefer_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) or "from .. import oracle" in s:
                    bad.append(f)
    assert not bad, bad


def test_ops_fail_loudly_on_cpu_tensors(lib):
    import pytest
    import torch
    from instancerefer_amd.sparse import SparseTensor, nn as spnn
    conv = spnn.Conv3d(4, 16, 3)
    st = SparseTensor(torch.randn(5, 4), torch.zeros(5, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="HIP device"):
        conv(st)


def test_cpp_autograd_nodes_module_builds_loads_and_binds_without_a_gpu():
    """csrc/_irx_nodes.so (csrc/torch_nodes.cpp: C++ autograd nodes over the C-ABI) is built in-tree, imports on a GPU-less host,
    binds the entry points of the library instance _lib.py loaded, and refuses to bind when one is missing. (The nodes' arithmetic
    is the C-ABI's: covered by the -m gpu operator tests.)"""
    from instancerefer_amd import _build, _lib, _nodes
    path = _build.build_nodes()
    assert os.path.exists(path)
    mod = _nodes.load()
    assert mod is not None and hasattr(mod, "mlp2") and hasattr(mod, "bind")
    lib = _lib.load()
    for name in _nodes._ENTRY_POINTS:
        assert hasattr(lib, name), name
    import pytest
    with pytest.raises(Exception):
        mod.bind({"irx_mlp2_fwd": 1})                 # incomplete table
    import ctypes
    mod.bind({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in _nodes._ENTRY_POINTS})   # restore
