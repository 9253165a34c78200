cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o ps -- python $GRAFT_REPO_ROOT/bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > /tmp/ps.log 2>&1
f=$(find /tmp/ps -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/stress; cp $f $GRAFT_REPO_ROOT/gpurun_out/stress/kernel_stats.csv
head -22 $f | cut -c1-150
