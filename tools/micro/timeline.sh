# Dev: one-step GPU timeline (fp32) from a rocprofv3 kernel trace -> gpurun_out/timeline/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/timeline; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-dtype --profile-steps 0 ${TL_ARGS} > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_overlap.py $f 3 > $O/overlap${TL_TAG}.txt 2>&1
tail -3 /tmp/tl.log | cut -c1-300
cat $O/overlap${TL_TAG}.txt
