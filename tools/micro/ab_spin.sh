# dev: host-only busy-wait per step -> does the step time follow it (host-paced) or not (GPU-paced)?
for d in f32 bf16; do for us in 0 300 600 1000; do
echo -n "$d spin=$us us: "; IRX_BENCH_SPIN_US=$us python bench.py --dtype $d --no-cpu-baseline --no-alt-dtype --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
