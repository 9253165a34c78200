V=$1; shift
for rep in 1 2 3; do for x in "$@"; do
env $V=$x timeout 300 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-alt-dtype --profile-steps 0 ${AB_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('$V=$x', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
