#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16_points.py -q -m gpu -s > gpurun_out/t_bf16pts.log 2>&1; echo "bf16 points rc=$?"; grep "HIP bf16 vs\|passed\|failed\|Error\|assert" gpurun_out/t_bf16pts.log | head -20; grep "differ" gpurun_out/t_bf16pts.log | sort -k3 -n -r | head -12
