#!/bin/bash
# GPU box: weight gradients one block behind (simq_tune_wgrad_overlap 4) and the three forwards side by side (simq_tune_fwd_overlap 2) --
# bit-identity checks on a deterministic plan, the parity tests of the step, then alternating A-B runs
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tests/diag/overlap_check.py 2>&1 | tail -16
python -m pytest tests/test_gpu_fcn.py tests/test_gpu_bnfuse.py tests/test_gpu_sized.py tests/test_gpu_bf16_points.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2 3 4; do
  for on in 1 4; do echo -n "fp32 configs1 fwd_overlap=2 wgrad_overlap=$on  "; bash tools/bv.sh --wgrad-overlap $on; done
done
for rep in 1 2; do
  for on in 0 2; do echo -n "fp32 configs1 fwd_overlap=$on  "; bash tools/bv.sh --fwd-overlap $on; done
  for on in 0 2; do echo -n "bf16 configs2 fwd_overlap=$on  "; bash tools/bv.sh --workload configs2 --fwd-overlap $on; done
done
for on in 0 2; do echo -n "fp32 configs3 fwd_overlap=$on  "; bash tools/bv.sh --workload configs3 --fwd-overlap $on --steps 10; done
