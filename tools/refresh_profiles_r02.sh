# Round-2 profile refresh on the GPU box: benches in every mode, rocprofv3 kernel stats (fp32 / bf16), PMC traffic passes.
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles_r02.sh <tag>'      -> gpurun_out/r02<tag>/
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02$1
mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline > $O/bench_bf16_b32.json 2>/dev/null
timeout 300 python bench.py --dtype bf16op --no-cpu-baseline > $O/bench_bf16op.json 2>/dev/null
timeout 300 python bench.py --workload attr --no-cpu-baseline > $O/bench_attr.json 2>/dev/null
timeout 400 python bench.py --workload stress --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_stress.json 2>/dev/null
timeout 300 python tools/op_count.py > $O/op_count.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o re -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > /tmp/pe.log 2>&1
cp /tmp/pe/*/re_kernel_stats.csv /tmp/pe/re_kernel_stats.csv 2>/dev/null; cp /tmp/pe/*/re_domain_stats.csv /tmp/pe/re_domain_stats.csv 2>/dev/null
cp /tmp/pe/re_kernel_stats.csv $GRAFT_REPO_ROOT/$O/kernel_stats.csv; cp /tmp/pe/re_domain_stats.csv $GRAFT_REPO_ROOT/$O/domain_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline > /tmp/pb.log 2>&1
cp /tmp/pb/*/rb_kernel_stats.csv /tmp/pb/rb_kernel_stats.csv 2>/dev/null
cp /tmp/pb/rb_kernel_stats.csv $GRAFT_REPO_ROOT/$O/bf16_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pf -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-alt-dtype > /tmp/pf.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pw -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-alt-dtype > /tmp/pw.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) $GRAFT_REPO_ROOT/$O/pmc_traffic.json
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pfb -o pf -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --dtype bf16 > /tmp/pfb.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pwb -o pw -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --dtype bf16 > /tmp/pwb.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $(find /tmp/pfb -name "*counter_collection.csv" | head -1) $(find /tmp/pwb -name "*counter_collection.csv" | head -1) $GRAFT_REPO_ROOT/$O/pmc_traffic_bf16.json
tail -2 /tmp/pf.log
ls $GRAFT_REPO_ROOT/$O
