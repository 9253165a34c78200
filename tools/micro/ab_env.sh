# usage: ab_env.sh VAR v1 v2 ... : default bench (quick) under VAR=value, twice each
V=$1; shift
for rep in 1 2; do for x in "$@"; do
env $V=$x timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-dtype --profile-steps 0 ${AB_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
