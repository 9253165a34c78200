# dev: host-only busy-wait per step -> does the step time follow it (host-paced) or not (GPU-paced)?   ab_spin.sh [bench args]
for us in 0 300 600 1000; do
echo -n "$* spin=$us us: "; IRX_BENCH_SPIN_US=$us python bench.py "$@" --no-cpu-baseline --no-alt-dtype --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done
