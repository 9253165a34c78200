for rep in 1 2; do for g in none freeze; do
IRX_BENCH_GC=$g timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('gc=$g fp32', round(d['value'],1), 'alt bf16', round(d['alt_dtype']['value'],1))"
done; done
