#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
  echo -n "rep $rep fp32-mfma  "; bash tools/bv.sh
  echo -n "rep $rep split3     "; bash tools/bv.sh --plan-option gemm_split=1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_split3.log
python -m pytest -q -m gpu --durations=12 --plan-option gemm_split=1 tests/test_gpu_fcn.py tests/test_gpu_fullsize.py tests/test_gpu_sized.py tests/test_gpu_bnfuse.py -k "not bf16" > gpurun_out/t_split.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/t_split.log | tail -40
