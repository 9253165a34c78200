#!/bin/bash
# Dev: time k_spconv3 variants built by build_abl3.sh with tools/conv3_bench.py.  Usage: run_abl3.sh tag1 tag2 ...
mkdir -p gpurun_out/abl3
for t in "$@"; do
echo "== $t"; IRX_LIB_PATH=tools/micro/libirx_s3_$t.so CHECK=0 ONLY="${ONLY:-stride 4,stride 2 3^3 fwd,stride 8 3^3 fwd}" timeout 300 python tools/conv3_bench.py 16 20 2>&1 | grep "n_out"
done > gpurun_out/abl3/out.txt 2>&1
cat gpurun_out/abl3/out.txt
