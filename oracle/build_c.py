"""Build the oracle's C/OpenMP port (oracle/csrc/spconv_cpu.c -> oracle/_ref/libirx_oracle_cpu.so).

Compiled for x86-64-v3 (AVX2 + FMA) so the prebuilt library runs on the GPU box's host CPU whatever built it.
The reference itself is pure Python, so there is no reference source to compile into oracle/_ref; this is
the plain-C restatement used as the CPU baseline and cross-checked against oracle/torchsparse. TEST INFRASTRUCTURE."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "spconv_cpu.c")
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libirx_oracle_cpu.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-ffast-math", "-shared", "-fPIC", SRC, "-o", tmp, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
