#!/bin/bash
# GPU box: A/B of the whole-planes-per-XCD walk of the fp32 batched transform-domain GEMM (simq_tune_plane_xcd)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "plane_per_xcd or winograd" 2>&1 | tail -5
for rep in 1 2 3; do
  for on in 0 1; do
    echo -n "plane_xcd=$on  "; bash tools/bv.sh --plane-xcd $on
  done
done
for on in 0 1; do
  python bench.py --no-cpu-baseline --no-extras --sustained-seconds 0 --steps 20 --warmup 5 --plane-xcd $on 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('plane_xcd=$on roofline', r.get('kernel'), r['achieved'], r['frac'], 'us', r.get('avg_launch_ms'), r.get('kernel_ms_per_step'))"
done
python tests/diag/diag_b128_bf16_loss.py 2>&1 | grep -v amdgpu.ids
SIMQ_LIBRARY=$PWD/spatial-intention-maps_amd/simq/libsimq_ablate.so SIMQ_BF16_IMG_HALF=0 python tests/diag/diag_b128_bf16_loss.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn"
