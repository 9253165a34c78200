# kernel trace of the bf16 step (real layout): overlap summary + per-kernel dump of one step.  bash tools/r05_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05$1; mkdir -p $O
rm -rf /tmp/pt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o rt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 ${BENCH_ARGS} > /tmp/pt.log 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_overlap.py $f 3 > $O/overlap.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_step_dump.py $f 3 > $O/step_dump.txt 2>&1
python $GRAFT_REPO_ROOT/tools/stats_groups.py $O/kernel_stats.csv 15 > $O/groups.txt
head -12 $O/overlap.txt; tail -1 /tmp/pt.log | cut -c1-160
