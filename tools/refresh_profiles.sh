set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01e
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r01e/pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r01e/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/r01e/bench.json 2> gpurun_out/r01e/bench.err
timeout 300 python bench.py --workload attr --no-cpu-baseline > gpurun_out/r01e/bench_attr.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o re -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/pe.log 2>&1
cp /tmp/pe/*stats* $GRAFT_REPO_ROOT/gpurun_out/r01e/
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pf -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline > /tmp/pf.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pw -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline > /tmp/pw.log 2>&1
ls /tmp/pf /tmp/pw
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pf/pf_counter_collection.csv /tmp/pw/pw_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/r01e/pmc_traffic.json
tail -3 /tmp/pf.log
