# serial (one stream) kernel stats of the bf16 step: alone-time of every kernel, per step.  bash tools/r05_serial.sh <tag>
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05$1; mkdir -p $O
rm -rf /tmp/ps
IRX_BENCH_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o rs -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 ${BENCH_ARGS} > /tmp/ps.log 2>&1
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
python $GRAFT_REPO_ROOT/tools/stats_groups.py $O/kernel_stats_serial.csv 45 | tee $O/serial_groups.txt
tail -1 /tmp/ps.log | cut -c1-200
