#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bnfuse.py -m gpu -q -x -k "wgrad or determin" 2>&1 | tail -3
for rep in 1 2 3; do
  for on in 0 1; do echo -n "fp32 wgrad_xcd_group=$on  "; bash tools/bv.sh --wgrad-xcd-group $on; done
done
for rep in 1 2 3; do
  for on in 0 1; do echo -n "bf16 wgrad_xcd_group=$on  "; bash tools/bv.sh --workload configs2 --wgrad-xcd-group $on; done
done
bash tools/r04_traffic.sh bf16_b128 --workload configs2 2>&1 | grep -i "wgrad\|total\|{" | cut -c1-400
