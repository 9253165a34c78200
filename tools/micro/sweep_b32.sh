run() { echo -n "$1 => "; env $1 timeout 300 python bench.py --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; }
run X=0
run IRX_SPCONV3_SPLIT_BELOW=256
run IRX_SPCONV3_SPLIT_BELOW=192
run IRX_SPCONV3_SPLIT_TARGET=256
run IRX_WGRAD3_XCD_MIN=100000
run IRX_WGRAD3_UNITS=384
run IRX_PAIRS_BUDGET=512
run X=1
