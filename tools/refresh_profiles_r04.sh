# Round-4 profile refresh on the GPU box: default bench (bf16 headline, fp32 alt, live PMC traffic), strong-scaling anchors at N = 1
# (B = 32 / 64 / 128 per GPU: the 32x4, 64x2 and 128x1 points of SURVEY 8(d) config 4), attr / stress benches, per-layer conv tables,
# rocprofv3 kernel stats of the bf16 and fp32 steps.   gpurun --timeout 3000 -- 'bash tools/refresh_profiles_r04.sh <tag>'
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04$1
mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
for b in 32 64 128; do
  timeout 400 python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-e2e > $O/bench_b$b.json 2>/dev/null
done
timeout 300 python bench.py --workload attr --no-cpu-baseline > $O/bench_attr.json 2>/dev/null
timeout 400 python bench.py --workload stress --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_stress.json 2>/dev/null
IRX_BENCH_LAYERS=1 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype > /dev/null 2> $O/conv_layers_bf16.txt
IRX_BENCH_LAYERS=1 timeout 300 python bench.py --dtype f32 --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype > /dev/null 2> $O/conv_layers_f32.txt
IRX_BENCH_CPROFILE=1 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-alt-dtype --profile-steps 0 > /dev/null 2> $O/host_cprofile.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > /tmp/pb.log 2>&1
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_bf16.csv; cp $(find /tmp/pb -name "*domain_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/domain_stats_bf16.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o re -- python $GRAFT_REPO_ROOT/bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > /tmp/pe.log 2>&1
cp $(find /tmp/pe -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_f32.csv
ls -la $GRAFT_REPO_ROOT/$O
