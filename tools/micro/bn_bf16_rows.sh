cd /tmp && export TMPDIR=/tmp
for r in ${BN_ROWS_LIST:-0 128 256}; do
rm -rf /tmp/pq; IRX_BN_ROWS=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o pq -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --dtype ${BN_DT:-bf16} --no-cpu-baseline --no-alt-dtype --profile-steps 0 > /tmp/pq.log 2>&1
f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1)
echo "IRX_BN_ROWS=$r"; python - "$f" <<'PY'
import csv,sys,re
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'\(.*','',r['Name'].replace('void ',''))
    if n.startswith('k_bn_'):
        tot+=int(r['TotalDurationNs'])
        print("   %-28s calls %5s avg %6.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3))
print("   BatchNorm total per step: %.3f ms" % (tot/1e6/11))
PY
done
