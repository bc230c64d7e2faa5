#!/bin/bash
# GPU box: the data-parallel / overlap tests after the three-forward form joined the communicator step; upload-stream A-B
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_dp.py tests/test_gpu_overlap.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2 3; do
  echo -n "fp32 configs1 upload on the consuming stream  "; bash tools/bv.sh --no-upload-stream
  echo -n "fp32 configs1 upload stream                   "; bash tools/bv.sh
done
for rep in 1 2; do
  echo -n "bf16 configs2 upload on the consuming stream  "; bash tools/bv.sh --workload configs2 --no-upload-stream
  echo -n "bf16 configs2 upload stream                   "; bash tools/bv.sh --workload configs2
done
python bench.py --no-cpu-baseline --no-extras --sustained-seconds 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('roofline (kernel alone)', r['achieved'], r['frac'], r['avg_launch_ms'], 'all tiles', r['all_gemm_tiles_frac'], 'whole step', r['whole_step_executed_frac'], 'M2', d['value'], 'M1', d['value_fwd_bwd_only'])"
