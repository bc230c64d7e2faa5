#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bnfuse.py -q -m gpu -s -k "bn_relu" > gpurun_out/t_bnkern.log 2>&1; echo "bn kernels rc=$?"; grep -c "one rounding" gpurun_out/t_bnkern.log; grep "one rounding" gpurun_out/t_bnkern.log | sort | uniq | head -40; tail -15 gpurun_out/t_bnkern.log
python -m pytest tests/test_gpu_bf16_points.py -q -m gpu -s -k "live" > gpurun_out/t_bf16pts.log 2>&1; echo "bf16 points rc=$?"; tail -25 gpurun_out/t_bf16pts.log
python tests/diag/diag_determinism.py 2>&1 | tail -20
