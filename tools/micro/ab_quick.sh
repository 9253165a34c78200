# dev: isolated conv kernels + three default bench runs
python tools/conv_microbench.py 16 2>/dev/null | grep stride
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-alt-dtype --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
