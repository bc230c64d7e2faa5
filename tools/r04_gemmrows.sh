#!/bin/bash
# GPU box: row tile of the batched transform-domain GEMM (ablation build) with the whole-planes-per-XCD walk in place
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for t in 64 96 128; do echo -n "fp32 ablate tile_rows=$t  "; SIMQ_GEMM_BATCHED_TILE=$t python tools/ab_step.py configs1 60 2>&1 | tail -1; done
done
