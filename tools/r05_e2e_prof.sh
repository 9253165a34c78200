cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05e2e; mkdir -p $O
rm -rf /tmp/pe2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe2 -o r -- python $GRAFT_REPO_ROOT/tools/e2e_train_bench.py --dtype bf16 --steps 20 --warmup 5 --scans 16 > /tmp/pe2.log 2>&1
cp $(find /tmp/pe2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_e2e.csv
tail -3 /tmp/pe2.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_e2e.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    print("%-90s %7s calls %8.1f us avg %8.2f ms total"%(r['Name'].replace('void ','')[:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
