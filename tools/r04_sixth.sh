#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tests/diag/diag_bf16_points.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16_points.py -q -m gpu -s > gpurun_out/t_bf16pts.log 2>&1; echo "bf16 points rc=$?"; grep "HIP bf16 vs\|the model\|full Q\|passed\|failed" gpurun_out/t_bf16pts.log
python -m pytest tests/test_gpu_bnfuse.py -q -m gpu -s -k deterministic > gpurun_out/t_det.log 2>&1; echo "det rc=$?"; grep "deterministic vs\|passed\|failed\|Error" gpurun_out/t_det.log | head
python tests/diag/diag_determinism.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
