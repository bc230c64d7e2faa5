#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/recipes/overlap.sh c3_b64_seq --workload configs3 --batch 64 --group-streams 0
bash tools/recipes/overlap.sh c3_b64_defer --workload configs3 --batch 64
bash tools/recipes/overlap.sh c3_b256_defer --workload configs3 --steps 4 --warmup 2
bash tools/recipes/overlap.sh c1 
for q in 2 3 4; do
  export GPU_MAX_HW_QUEUES=$q
  echo -n "hwq $q configs1         "; bash tools/bv.sh
  echo -n "hwq $q configs2         "; bash tools/bv.sh --workload configs2
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_hwq_small.log
