#!/bin/bash
# Dev: time the k_spconv2 ablation variants (tools/micro/build_abl.sh) with tools/conv_microbench.py
mkdir -p gpurun_out/abl
for m in "$@"; do for dt in fp32 bf16_operands; do
echo "== ABL=$m $dt"; IRX_LIB_PATH=tools/micro/libirx_abl$m.so IRX_DTYPE=$dt timeout 300 python tools/conv_microbench.py 2>&1 | grep stride
done; done > gpurun_out/abl/out.txt 2>&1
cat gpurun_out/abl/out.txt
