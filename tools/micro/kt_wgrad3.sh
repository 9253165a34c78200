# Dev: per-kernel GPU durations (rocprofv3 --kernel-trace, no counters) of tools/wgrad3_bench.py: the python loop of that tool
# is host-bound below ~30 us per call, the trace is not.  Usage: tools/micro/kt_wgrad3.sh [tag]   (knob env vars are inherited)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt_wgrad3; mkdir -p $O
TAG=${1:-default}
export CHECK=0
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/wgrad3_bench.py 16 10 > /tmp/kt.log 2>&1
python - <<P > $O/$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_wgrad3' in k or 'k_wgrad_pairs' in k or 'k_pairs_reduce' in k:
            acc[(k, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(acc):
    v = sorted(acc[k])
    print('%-60s grid %8s %4s  n %3d  median %7.1f us  min %7.1f' % (k[0][:60], k[1], k[2], len(v), v[len(v) // 2], v[0]))
P
cat $O/$TAG.txt
