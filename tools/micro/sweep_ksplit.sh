run() { echo -n "$1 $2 => "; env $1 timeout 300 python bench.py $2 --steps 40 --warmup 10 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2; do
run X=0 "--batch 32"
run IRX_SPCONV3_KSPLIT=1 "--batch 32"
run IRX_SPCONV3_SPLIT_BELOW=128 "--batch 32"
done
run X=0 "--batch 64"
run IRX_SPCONV3_KSPLIT=1 "--batch 64"
run X=0 "--batch 8"
run IRX_SPCONV3_KSPLIT=1 "--batch 8"
