# dev, TIMING ONLY: the training step with one BatchNorm pass of the encoders removed (IRX_BN_ABL, results wrong) = the most a
# perfect fusion of that pass into a neighbouring kernel could buy
for i in 1 2; do for m in 0 1 2 4 8 15; do
echo -n "IRX_BN_ABL=$m: "; IRX_BN_ABL=$m python bench.py --no-cpu-baseline --no-alt-dtype --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done; done
