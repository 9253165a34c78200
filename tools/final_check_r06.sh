# Round 6, last GPU session: suite, smoke, the default bench line, the sort microbench and the two end-to-end A/Bs in one call.
#   gpurun --timeout 2400 -- 'bash tools/final_check_r06.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_d; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -q -x --durations=15 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 200 python tools/micro/sort_bench.py > $O/sort_bench.txt 2>&1; tail -5 $O/sort_bench.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json | head -c 300; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-e2e > $O/bench_driver_sized.json 2>/dev/null
for v in "" "IRX_INPUT_VOXELIZE_LAUNCH=1" "IRX_E2E_POST=early" "IRX_INPUT_VOXELIZE_LAUNCH=1 IRX_E2E_POST=early"; do
  echo "== e2e $v"
  env $v IRX_E2E_TIMES=1 timeout 300 python tools/e2e_train_bench.py --dtype bf16 --steps 200 --warmup 150 2>&1 | grep "worker:\|end to end\|same process\|Error\|error" | tee -a $O/e2e_ab.txt
done
