#!/bin/bash
# GPU box: LDS / wave-stall counters of the image-tile forward kernel (tools/pp_check.py) and the image-tile weight gradient
# (tools/wgrad_check.py) at B = 128
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lds_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local tag=$1; local tool=$2; shift; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O -o ${tag} -- python $R/tools/$tool 128 > $O/${tag}.out 2>&1; }
run fwd_lds pp_check.py SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run fwd_sq pp_check.py SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run wg_lds wgrad_check.py SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run wg_sq wgrad_check.py SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
cd $R
for n in fwd_lds fwd_sq wg_lds wg_sq; do
  echo "== $n"; python tools/rocprof_summary.py $(find $O -name "${n}_results.db") 2>/dev/null | grep -A400 "PMC counters" | grep "igemm_bf16_img\|wgrad_bf16_img"
done > $O/summary.txt
find $O -name "*.db" -delete
cat $O/summary.txt
