# Dev: SQ counters of the isolated conv kernels (tools/conv_microbench.py), two --pmc passes -> gpurun_out/pmc_conv/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d /tmp/p1 -o p1 -- python $GRAFT_REPO_ROOT/tools/conv_microbench.py 16 > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/p2 -o p2 -- python $GRAFT_REPO_ROOT/tools/conv_microbench.py 16 > /tmp/p2.log 2>&1
tail -2 /tmp/p1.log /tmp/p2.log | cut -c1-200
python $GRAFT_REPO_ROOT/tools/pmc_sq_fold.py $O/sq_counters.txt $(find /tmp/p1 /tmp/p2 -name "*counter_collection.csv")
