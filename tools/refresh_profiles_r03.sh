# Round-3 profile refresh on the GPU box: default bench (live PMC traffic inside), bf16 / attr / stress benches, rocprofv3 kernel
# stats for the fp32 and bf16 steps.   gpurun --timeout 2400 -- 'bash tools/refresh_profiles_r03.sh <tag>'  -> gpurun_out/r03<tag>/
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03$1
mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline > $O/bench_bf16_b32.json 2>/dev/null
timeout 300 python bench.py --workload attr --no-cpu-baseline > $O/bench_attr.json 2>/dev/null
timeout 400 python bench.py --workload stress --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_stress.json 2>/dev/null
IRX_BENCH_LAYERS=1 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype > /dev/null 2> $O/conv_layers.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o re -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > /tmp/pe.log 2>&1
cp $(find /tmp/pe -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats.csv; cp $(find /tmp/pe -name "*domain_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/domain_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline > /tmp/pb.log 2>&1
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/bf16_kernel_stats.csv
ls -la $GRAFT_REPO_ROOT/$O
