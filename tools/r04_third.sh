#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bnfuse.py -q -m gpu -x > gpurun_out/t_bnfuse.log 2>&1; echo "bnfuse rc=$?"; tail -3 gpurun_out/t_bnfuse.log
python tools/bnfuse_check.py 2>&1 | grep "B= 32" 
bash tools/bv.sh; bash tools/bv.sh --workload configs2
bash tools/kt.sh r04_b32 > /dev/null
grep -n "wino\|bn_\|conv_img" gpurun_out/r04_b32_kernel_trace.txt | cut -c1-130
