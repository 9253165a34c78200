# k_stem_wgrad + its reduce vs the number of partial blocks (IRX_STEM_WGRAD_BLOCKS): rocprofv3 kernel stats of the bf16 step
cd /tmp && export TMPDIR=/tmp
for b in ${BLOCKS:-1024 512 256}; do
rm -rf /tmp/pq; IRX_STEM_WGRAD_BLOCKS=$b IRX_BENCH_PRIME=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o rq -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /tmp/pq.log 2>&1
echo "== blocks cap $b"; python - $(find /tmp/pq -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_wgrad_reduce" in r["Name"] or "k_stem_wgrad" in r["Name"]: print("  ", r["Name"][:50], r["Calls"], "avg", round(float(r["AverageNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1), "total/step", round(float(r["TotalDurationNs"])/15e3,1))
PY
done
