# Dev: L2 / fabric counters of the isolated bf16 weight-gradient kernel (tools/wgrad3_bench.py), separate --pmc passes
# -> gpurun_out/pmc_wgrad3/.  Usage: tools/micro/pmc_wgrad3.sh ["ONLY filter"]   (IRX_PAIRS_BUDGET etc. are inherited)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad3; mkdir -p $O
export CHECK=0 ONLY="${1:-stride 4 3^3,stride 2 3^3}"
TAG=${TAG:-default}
run() { n=$1; shift; rm -rf /tmp/w$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/w$n -o w$n -- python $GRAFT_REPO_ROOT/tools/wgrad3_bench.py 16 5 > /tmp/w$n.log 2>&1; tail -2 /tmp/w$n.log | cut -c1-160; }
run 1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run 2 FETCH_SIZE WRITE_SIZE
run 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
python - <<P > $O/$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/w[123]/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_wgrad3' in k or 'k_wgrad_pairs' in k or 'k_pairs_reduce' in k:
            acc[k + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('   %-28s mean %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
P
cat $O/$TAG.txt
