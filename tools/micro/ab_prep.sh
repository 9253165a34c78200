# Dev A/B: where the encoders' plans / kernel maps are built (forward vs preparation stage, inline vs helper thread)
run() { echo -n "$1 | $2 => "; env $1 timeout 300 python bench.py $2 --steps 60 --warmup 10 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2; do
run X=0 ""
run IRX_PREP_PLANS=1 ""
run X=0 "--prep-thread"
run IRX_PREP_PLANS=1 "--prep-thread"
done
