# Round-5 measurement bundle: driver-like bench (5 warm-up / 20 steps), long bench, main-stream timeline, kernel trace of the bf16 step
# (per-queue busy / gaps + launch counts).   gpurun --timeout 1500 -- 'bash tools/r05_baseline.sh <tag>'
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05$1
mkdir -p $O
nproc > $O/host.txt; lscpu | grep "Model name" >> $O/host.txt
timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} > $O/bench_driver.json 2> $O/bench_driver.err
timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-alt-dtype --no-e2e ${BENCH_ARGS} > $O/bench_long.json 2>/dev/null
IRX_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 ${BENCH_ARGS} > /dev/null 2> $O/timeline_bf16.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 ${BENCH_ARGS} > /tmp/pb.log 2>&1
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16.csv
f=$(find /tmp/pb -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_overlap.py $f 3 > $O/overlap_bf16.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_step_dump.py $f 3 > $O/step_dump_bf16.txt 2>&1
tail -2 /tmp/pb.log | cut -c1-400
cut -c1-600 $O/bench_driver.json; cut -c1-300 $O/bench_long.json
