# per-kernel durations of one BatchNorm forward size: bash tools/micro/bn_one.sh "n c bf" ...
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
rm -rf /tmp/pz; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pz -o rz -- python $GRAFT_REPO_ROOT/tools/micro/bn_one.py $spec > /tmp/pz.log 2>&1
f=$(find /tmp/pz -name "*kernel_stats.csv" | head -1)
echo "== $spec"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_bn_' in r['Name']: print("   %-70s calls %4s avg %6.1f us"%(r['Name'][:70].replace('void ',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
done
