#!/bin/bash
# GPU box: PMC passes over tools/pp_check.py for the ping-pong bf16 kernel (L2 hit rate, fabric traffic, wave stall buckets)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pp_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc "$@" -d $O -o $1 -- python $R/tools/pp_check.py 128 > $O/$1.out 2>&1; }
run TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run FETCH_SIZE
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
cd $R
for n in TCC_HIT_sum FETCH_SIZE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS; do
  python tools/rocprof_summary.py $(find $O -name "${n}_results.db") 2>/dev/null | grep -A400 "PMC counters" | grep "igemm_bf16_pp\|igemm_bf16_dma" 
done > $O/summary.txt
python tools/rocprof_summary.py $(find $O -name "TCC_HIT_sum_results.db") | head -12 >> $O/summary.txt
find $O -name "*.db" -delete
cat $O/summary.txt
