#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "pingpong_kernels or lds_dma" > gpurun_out/t_half.log 2>&1; echo "half-map op tests rc=$?"; tail -3 gpurun_out/t_half.log
python tools/img_half_check.py 2>&1 | grep -v amdgpu.ids
bash tools/bv.sh --workload configs2
python -m pytest tests/test_gpu_fcn.py tests/test_gpu_fullsize.py tests/test_gpu_intention.py tests/test_gpu_ops.py tests/test_gpu_sized.py -q -m gpu -s -x --durations=10 > gpurun_out/t_rest.log 2>&1; echo "rest rc=$?"
grep -n "passed\|failed\|^FAILED" gpurun_out/t_rest.log | tail -8
grep -n "medians" gpurun_out/t_rest.log | tail -8
grep -n "gs_b128" gpurun_out/t_rest.log | tail -14
