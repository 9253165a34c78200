cd /tmp && export TMPDIR=/tmp
for r in ${ROWS:-256 128}; do
rm -rf /tmp/pq; IRX_BENCH_PRIME=0 IRX_BN_ROWS_MIN=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o rq -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /tmp/pq.log 2>&1
f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1)
echo "== rows min $r"; python - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'k_bn_' in n:
        t=float(r['TotalDurationNs'])/15/1e3; tot+=t
        if any(k in n for k in ('slabs','partial<','finalize')): print("   %-60s %5.1f/step %7.1f us/step avg %5.1f"%(n[:60].replace('void ',''), int(r['Calls'])/15, t, float(r['AverageNs'])/1e3))
print("   BN total us/step", round(tot,1))
PY
done
