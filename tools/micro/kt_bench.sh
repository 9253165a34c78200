# Dev: rocprofv3 kernel trace of a short bench run, per (kernel, grid) durations for kernels matching $1 (regex, default k_bn)
cd /tmp && export TMPDIR=/tmp
PAT=${1:-k_bn}
rm -rf /tmp/ktb; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktb -o ktb -- python $GRAFT_REPO_ROOT/bench.py ${BENCH_ARGS:---steps 10 --warmup 3} --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 > /tmp/ktb.log 2>&1
python - "$PAT" <<'P'
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/ktb/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if pat.search(k):
            acc[(k.split('(')[0][:70], int(r.get('Grid_Size_X', r.get('Grid_Size', 0))), int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print('%-72s grid %8d wg %4d  n %4d  median %7.1f us  total %8.1f us (%4.1f %%)' % (k[0], k[1], k[2], len(v), v[len(v) // 2], sum(v), 100 * sum(v) / tot))
P
