#!/bin/bash
# GPU box: weight gradients one block behind (simq_tune_wgrad_overlap 4) -- bit-identity check on a deterministic plan, then alternating A-B
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/wov_check.py 2>&1 | tail -12
for rep in 1 2 3; do
  for on in 1 4; do echo -n "fp32 configs1 wgrad_overlap=$on  "; bash tools/bv.sh --wgrad-overlap $on; done
done
for on in 1 4; do echo -n "fp32 configs3 wgrad_overlap=$on  "; bash tools/bv.sh --workload configs3 --wgrad-overlap $on --steps 10; done
# where the no-grad forwards are forked (simq_tune_fwd_overlap; 2 = timing only)
for rep in 1 2; do
  for on in 0 1 2; do echo -n "fp32 configs1 fwd_overlap=$on  "; bash tools/bv.sh --fwd-overlap $on; done
done
echo -n "fp32 configs1 no-overlap  "; bash tools/bv.sh --no-overlap
for on in 0 2; do echo -n "bf16 configs2 fwd_overlap=$on  "; bash tools/bv.sh --workload configs2 --fwd-overlap $on; done
