#!/bin/bash
# GPU box: the overlap tests, then M1 / M2 with the library-owned side stream of the standalone backward (A-B on one box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_overlap.py -m gpu -q -x 2>&1 | tail -6
for rep in 1 2 3; do
  for on in 0 4; do echo -n "fp32 configs1 wgrad_overlap=$on  "; bash tools/bv.sh --wgrad-overlap $on; done
done
for on in 0 4; do echo -n "fp32 configs3 wgrad_overlap=$on  "; bash tools/bv.sh --workload configs3 --wgrad-overlap $on --steps 10; done
echo -n "bf16 configs2 default  "; bash tools/bv.sh --workload configs2
