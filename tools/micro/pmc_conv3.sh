# Dev: SQ / TA / TCP counters of the isolated third-generation conv kernel (tools/conv3_bench.py), separate --pmc passes
# -> gpurun_out/pmc_conv3/.  Usage: tools/micro/pmc_conv3.sh ["ONLY filter"]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv3; mkdir -p $O
export CHECK=0 ONLY="${1:-stride 4 3^3 fwd,stride 2 3^3 fwd}"
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/q$n -o q$n -- python $GRAFT_REPO_ROOT/tools/conv3_bench.py 16 5 > /tmp/q$n.log 2>&1; tail -2 /tmp/q$n.log | cut -c1-160; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES
run 2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES
run 3 TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run 4 TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
run 5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run 6 GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
rocprofv3 -L 2>/dev/null | grep -o "^[A-Za-z_0-9]*\|Name: *[A-Za-z_0-9]*" | sort -u > $O/counter_names.txt 2>/dev/null
python $GRAFT_REPO_ROOT/tools/pmc_sq_fold.py $O/sq_counters.txt $(find /tmp/q1 /tmp/q2 /tmp/q3 /tmp/q4 /tmp/q5 /tmp/q6 -name "*counter_collection.csv")
