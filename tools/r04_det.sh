#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for d in 0 1; do echo -n "fp32 deterministic=$d  "; bash tools/bv.sh --plan-option deterministic=$d; done
done
for rep in 1 2 3; do
  for d in 0 1; do echo -n "bf16 deterministic=$d  "; bash tools/bv.sh --workload configs2 --plan-option deterministic=$d; done
done
