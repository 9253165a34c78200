"""Build libirx.so (the C-ABI HIP library) in-tree for gfx950.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the build container
as well as on the GPU box. The .so is git-ignored but travels with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("IRX_LIB_PATH") or os.path.join(CSRC, "libirx.so")   # override: dev-only instrumented builds
SOURCES = ["irx_coords.hip", "irx_sort.hip", "irx_sched.hip", "irx_spconv.hip", "irx_spconv2.hip", "irx_spconv3.hip", "irx_pairs.hip", "irx_encoder.hip", "irx_stem.hip", "irx_norm.hip", "irx_pool.hip", "irx_match.hip", "irx_mlp.hip", "irx_input.hip", "irx_gru.hip", "irx_optim.hip", "irx_labels.hip", "irx_project.hip", "irx_edgeconv.hip"]
HEADERS = ["irx_common.h", os.path.join("..", "..", "include", "irx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-pthread"]
# Sources whose results must equal numpy / torch element-wise arithmetic bit for bit (float64 IoUs, the projection's
# float32 plane / pixel tests): no fused multiply-adds except the explicit ones. (-ffp-contract=fast fuses in the backend,
# so `#pragma clang fp contract(off)` inside the file does not stop it: measured, 24 v_fma_f64 vs 6.)
EXTRA_FLAGS = {"irx_labels.hip": ["-ffp-contract=off"], "irx_project.hip": ["-ffp-contract=off"]}


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link csrc/libirx.so. Returns the library path."""
    if os.environ.get("IRX_LIB_PATH"):
        return LIB_PATH                # a hand-built variant: never rebuilt here
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time (torchrun starts N ranks at once); late arrivals find a fresh library and return
    lock = open(os.path.join(objdir, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not _stale():
            return LIB_PATH
        return _build_locked(objdir, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(objdir, verbose):

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, LIB_PATH)          # atomic: a concurrent loader never sees a half-written library
    return LIB_PATH


# ---- C++ autograd nodes over the C-ABI (csrc/torch_nodes.cpp -> csrc/_irx_nodes.so) ----------------------------------------
# A Python extension module compiled against the torch headers with g++ (host code only: it calls libirx through function
# addresses and never touches HIP). In-tree like libirx.so, so it travels to the GPU box with the snapshot.
NODES_SRC = os.path.join(CSRC, "torch_nodes.cpp")
NODES_SRCS = [NODES_SRC, os.path.join(CSRC, "heads_nodes.cpp")]       # heads_nodes.cpp: one node per head (round 6)
NODES_HDRS = [os.path.join(CSRC, "torch_nodes.h")]
NODES_PATH = os.path.join(CSRC, "_irx_nodes.so")
CXX = os.environ.get("CXX", "g++")


def nodes_stale() -> bool:
    if not os.path.exists(NODES_PATH):
        return True
    t = os.path.getmtime(NODES_PATH)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in NODES_SRCS + NODES_HDRS)


def build_nodes(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/torch_nodes.cpp + csrc/heads_nodes.cpp into csrc/_irx_nodes.so (torch C++ extension, ~1 min; the two translation
    units compile in parallel). Returns the module path."""
    if not force and not nodes_stale():
        return NODES_PATH
    import fcntl
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    lock = open(os.path.join(objdir, ".lock_nodes"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not nodes_stale():
            return NODES_PATH
        tlib = ce.library_paths()[0]
        base = [CXX, "-O2", "-std=c++17", "-fPIC", "-DTORCH_EXTENSION_NAME=_irx_nodes", "-DTORCH_API_INCLUDE_EXTENSION_H",
                "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
        base += ["-I" + d for d in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]

        def compile_one(src):
            obj = os.path.join(objdir, os.path.basename(src).replace(".cpp", ".nodes.o"))
            cmd = base + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("building _irx_nodes failed (%s):\n%s\n%s" % (os.path.basename(src), r.stdout, r.stderr))
            return obj

        with ThreadPoolExecutor(max_workers=len(NODES_SRCS)) as ex:
            objs = list(ex.map(compile_one, NODES_SRCS))
        tmp = NODES_PATH + ".tmp.%d" % os.getpid()
        cmd = [CXX, "-shared", "-fPIC"] + objs + ["-o", tmp, "-L" + tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("linking _irx_nodes failed:\n%s\n%s" % (r.stdout, r.stderr))
        os.replace(tmp, NODES_PATH)
        return NODES_PATH
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_nodes(force="--force" in sys.argv, verbose="-v" in sys.argv))
