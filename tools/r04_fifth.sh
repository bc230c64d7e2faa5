#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bnfuse.py -q -m gpu -s -k "bn_relu" > gpurun_out/t_bnkern.log 2>&1; echo "bn kernels rc=$?"; grep "one rounding" gpurun_out/t_bnkern.log | sort | uniq | head -40; tail -5 gpurun_out/t_bnkern.log
python tests/diag/diag_bf16_points.py 2>&1 | grep -v amdgpu.ids
