# Dev: bf16 step with the third-generation conv on / off (knob IRX_SPCONV3), B = 16 and 32, plus the per-layer table.
# gpurun --timeout 1500 -- 'bash tools/ab_bf16.sh <tag>'  -> gpurun_out/ab_bf16_<tag>/
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_bf16_$1; mkdir -p $O
for v in 1 0; do
  IRX_SPCONV3=$v timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-e2e > $O/b16_v$v.json 2>/dev/null
  IRX_SPCONV3=$v timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-e2e > $O/b32_v$v.json 2>/dev/null
  IRX_SPCONV3=$v IRX_BENCH_LAYERS=1 timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 4 --no-cpu-baseline --no-e2e > /dev/null 2> $O/layers_v$v.txt
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, "%.0f scenes/s %.2f ms" % (d["value"], d["ms_per_step"]), r.get("kernel"), "frac %.3f avg_us %.1f" % (r.get("frac",0), r.get("avg_launch_us",0)))
    except Exception as e: print(f, "ERR", e)
PY
