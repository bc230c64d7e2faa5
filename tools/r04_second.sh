#!/bin/bash
# round 4: fused-BatchNorm kernels (tests + per-shape cost), kernel traces of both legs, A/B of the step with / without the fusion
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bnfuse.py -q -m gpu -x > gpurun_out/t_bnfuse.log 2>&1; echo "bnfuse rc=$?"; tail -3 gpurun_out/t_bnfuse.log
python tools/bnfuse_check.py > gpurun_out/bnfuse_check.txt 2>&1; cat gpurun_out/bnfuse_check.txt
bash tools/bv.sh; bash tools/bv.sh --workload configs2
bash tools/kt.sh r04_b32 > /dev/null
bash tools/kt.sh r04_bf16_b128 --workload configs2 > /dev/null
head -36 gpurun_out/r04_b32_kernel_trace.txt | cut -c1-130
grep -n "bn_\|igemm_bf16_img" gpurun_out/r04_bf16_b128_kernel_trace.txt | cut -c1-130
