#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16_points.py tests/test_gpu_ops.py -q -m gpu -x -k "dense_gradient or f32_pingpong or pingpong_kernels" > gpurun_out/t_fix.log 2>&1; echo "fix rc=$?"; tail -3 gpurun_out/t_fix.log
for i in 1 2; do bash tools/bv.sh; bash tools/bv.sh --no-overlap; done
for i in 1 2; do bash tools/bv.sh --workload configs2; bash tools/bv.sh --workload configs2 --no-overlap; done
