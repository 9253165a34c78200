# Round-6 profile refresh on the GPU box.   gpurun --timeout 3000 -- 'bash tools/refresh_profiles_r06.sh <tag>'
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06$1
mkdir -p $O
nproc > $O/host.txt; lscpu | grep "Model name" >> $O/host.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e > $O/bench_driver_sized.json 2>/dev/null
for b in 32 128; do
  timeout 400 python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-alt-dtype > $O/bench_b$b.json 2>/dev/null
done
timeout 300 python bench.py --workload attr --no-cpu-baseline > $O/bench_attr.json 2>/dev/null
timeout 400 python bench.py --workload stress --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_stress.json 2>/dev/null
IRX_BENCH_LAYERS=1 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-e2e > /dev/null 2> $O/conv_layers_bf16.txt
IRX_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /dev/null 2> $O/timeline_bf16.txt
IRX_BENCH_TIMELINE=1 timeout 300 python bench.py --dtype f32 --steps 20 --warmup 8 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /dev/null 2> $O/timeline_f32.txt
IRX_BENCH_TORCHPROF=1 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /dev/null 2> $O/host_torchprof_bf16.txt
IRX_BENCH_PREP_WORKER=all timeout 300 python bench.py --no-cpu-baseline --no-alt-dtype --no-e2e > $O/bench_prep_worker_all.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-alt-dtype --no-e2e > $O/bench_again.json 2>/dev/null
timeout 200 python tools/kmap_bench.py > $O/kmap_bench.txt 2>&1
# (IRX_BENCH_PRIME_S=0: exactly 30 primed + 5 warm-up + 10 timed = 45 steps in the traced process, the divisor of stats_groups.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb /tmp/pe
IRX_BENCH_PRIME_S=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /tmp/pb.log 2>&1
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_bf16.csv
T=$(find /tmp/pb -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/stats_groups.py $GRAFT_REPO_ROOT/$O/kernel_stats_bf16.csv 45 > $GRAFT_REPO_ROOT/$O/kernel_groups_bf16.txt
python $GRAFT_REPO_ROOT/tools/trace_overlap.py $T 3 > $GRAFT_REPO_ROOT/$O/trace_overlap.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_step_dump.py $T 3 > $GRAFT_REPO_ROOT/$O/trace_step_dump.txt 2>&1
IRX_BENCH_PRIME_S=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o re -- python $GRAFT_REPO_ROOT/bench.py --dtype f32 --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /tmp/pe.log 2>&1
cp $(find /tmp/pe -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_f32.csv
python $GRAFT_REPO_ROOT/tools/stats_groups.py $GRAFT_REPO_ROOT/$O/kernel_stats_f32.csv 45 > $GRAFT_REPO_ROOT/$O/kernel_groups_f32.txt
cd $GRAFT_REPO_ROOT
ls -la $O
