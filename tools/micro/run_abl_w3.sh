#!/bin/bash
# Dev: time k_wgrad3 variants built by build_abl_w3.sh with tools/wgrad3_bench.py.  Usage: run_abl_w3.sh tag1 tag2 ...
mkdir -p gpurun_out/ablw3
for t in "$@"; do
echo "== $t"; IRX_LIB_PATH=tools/micro/libirx_w3_$t.so CHECK=0 ONLY="${ONLY:-stride 4,stride 2 3^3,stride 8 3^3}" timeout 300 python tools/wgrad3_bench.py 16 20 2>&1 | grep "n_out"
done > gpurun_out/ablw3/out.txt 2>&1
cat gpurun_out/ablw3/out.txt
