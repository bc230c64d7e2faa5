#!/bin/bash
# the fp32 batched GEMM alone per shape (tools/gemm_batched_probe.py): product build, then the ablation build's N-tile runs
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
( echo "== product"; python tools/gemm_batched_probe.py 32 29
  for r in 2 4; do echo "== ablation build SIMQ_GEMM_NT_RUN=$r"; SIMQ_LIBRARY=$PWD/spatial-intention-maps_amd/simq/libsimq_ablate.so SIMQ_GEMM_NT_RUN=$r python tools/gemm_batched_probe.py 32 2>&1 | grep -v Warn; done
) 2>&1 | grep -v "amdgpu.ids\|warnings.warn" | tee gpurun_out/gemm_probe.log
python -m pytest -q -m gpu tests/test_gpu_bf16_points.py -k "backward_teacher" 2>&1 | tail -3
