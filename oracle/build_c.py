"""Build the oracle's C/OpenMP port (oracle/csrc/spconv_cpu.c) into oracle/_build/ — TEST INFRASTRUCTURE.

This is the BUILDER'S OWN plain-C restatement of the torchsparse CPU algorithm, not a compiled copy of the reference: the
reference is pure Python, so there is no reference source to compile and nothing here pins parity by itself — the C
port is pinned TO the Python oracle by tests/test_oracle_cpu.py. Two builds of the same source:
  libirx_oracle_cpu.so       strict IEEE arithmetic (no -ffast-math): what every parity test loads (cpu_port.lib());
  libirx_oracle_cpu_fast.so  -ffast-math: bench.py's timed `cpu_baseline` leg only (cpu_port.lib(fast=True)).
Compiled for x86-64-v3 (AVX2 + FMA) so the prebuilt libraries run on the GPU box's host CPU whatever built them."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "spconv_cpu.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libirx_oracle_cpu.so")
LIB_FAST = os.path.join(OUT_DIR, "libirx_oracle_cpu_fast.so")


def _build_one(out, extra, force):
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC):
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp"] + extra + ["-shared", "-fPIC", SRC, "-o", tmp, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + r.stderr)
    os.replace(tmp, out)
    return out


def build(force=False, fast=False):
    """-> path of the strict build (parity) or of the -ffast-math build (timed baseline). build(force=True) rebuilds both."""
    strict = _build_one(LIB, ["-ffp-contract=off"], force)
    if fast or force:
        quick = _build_one(LIB_FAST, ["-ffast-math"], force)
        if fast:
            return quick
    return strict


if __name__ == "__main__":
    print(build(force=True))
