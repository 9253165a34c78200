cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16_operands; do
rm -rf /tmp/pa; IRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/micro/libirx_abl$1.so IRX_DTYPE=$dt timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o pa -- python $GRAFT_REPO_ROOT/tools/conv_microbench.py > /tmp/pa.log 2>&1
f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
echo "== $dt"; python - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'\(.*','',r['Name'].replace('void ',''))
    if 'spconv2<' in n or 'permute' in n or 'wgrad_reduce' in n:
        print("   %-40s calls %5s avg %7.1f min %7.1f max %7.1f us" % (n[:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
