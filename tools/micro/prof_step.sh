# Dev: rocprofv3 kernel trace of a short bench run -> gpurun_out/<tag>_{kernel_stats.csv,overlap.txt,stepdump.txt,groups.txt}
#   bash tools/micro/prof_step.sh <tag> [extra bench.py arguments]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
IRX_BENCH_PRIME_S=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $R/bench.py --no-alt-dtype --no-cpu-baseline --no-e2e --profile-steps 0 --steps 40 --warmup 10 "$@" > $O/${tag}_prof.json 2> $O/${tag}_prof.err
T=$(find /tmp/pp -name "*kernel_trace.csv" | head -1); S=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
cd $R
python tools/trace_overlap.py $T 4 > $O/${tag}_overlap.txt 2>&1
python tools/trace_step_dump.py $T 4 > $O/${tag}_stepdump.txt 2>&1
cp $S $O/${tag}_kernel_stats.csv
python tools/stats_groups.py $S 80 > $O/${tag}_groups.txt 2>&1     # 30 primed + 10 warm-up + 40 timed steps
head -12 $O/${tag}_overlap.txt
