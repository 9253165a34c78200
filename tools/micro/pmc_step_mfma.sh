# Dev: matrix-core busy cycles of the whole training step (one --pmc pass, --kernel-trace only) -> gpurun_out/pmc_step/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_step; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/pm -o pm -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype --profile-steps 0 --no-pipeline > /tmp/pm.log 2>&1
python - <<'PY'
import csv, glob, collections, os
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']; n = n[5:] if n.startswith('void ') else n; n = n.split('(')[0]
    acc[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES': calls[n] += 1
steps = calls['k_adam'] or 8
tot = sum(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for v in acc.values())
out = ["steps %d; SQ_VALU_MFMA_BUSY_CYCLES summed over all kernels: %.4g per step" % (steps, tot / steps)]
for n, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0))[:14]:
    out.append("  %-44s %6.1f launches/step  mfma busy %.4g / step (%.1f %%)" % (n[:44], calls[n] / steps, v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / steps, 100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(tot, 1)))
open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_step/mfma_busy.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -c 300 /tmp/pm.log
