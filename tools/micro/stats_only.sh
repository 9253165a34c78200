cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o re -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > /tmp/pe.log 2>&1
cp $(find /tmp/pe -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; cp $(find /tmp/pe -name "*domain_stats.csv" | head -1) $O/domain_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline > /tmp/pb.log 2>&1
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $O/bf16_kernel_stats.csv
head -4 $O/kernel_stats.csv | cut -c1-160
