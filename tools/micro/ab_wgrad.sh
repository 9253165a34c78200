# dev A/B: first- vs second-generation fp32 pair-list weight-gradient (isolated launches, then the training step)
echo "--- gen2"; python tools/conv_microbench.py 16 2>/dev/null | grep stride
echo "--- gen1"; IRX_WGRAD_V1=1 python tools/conv_microbench.py 16 2>/dev/null | grep stride
for i in 1 2; do for v in 0 1; do
echo -n "fp32 step wgrad_v1=$v "; IRX_WGRAD_V1=$v python bench.py --no-cpu-baseline --no-alt-dtype --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
