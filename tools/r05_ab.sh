# A/B of env settings on one box, alternating, two rounds: bash tools/r05_ab.sh "K=V ..." "K=V ..."   (BENCH_ARGS extra args)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O
for rep in 1 2; do
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-alt-dtype --no-e2e --profile-steps 0 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $O/ab.txt
done
done
