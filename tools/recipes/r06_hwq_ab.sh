#!/bin/bash
# round 6: does the HIP runtime's default of 4 hardware queues per process serialise the step's independent streams?  (one learner uses
# launch + side + early + third + loss-copy + upload = 6 streams, two concurrent robot groups 11)  GPU_MAX_HW_QUEUES A/B on one box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
 for q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  echo -n "rep $rep hwq $q configs1         "; bash tools/bv.sh
  echo -n "rep $rep hwq $q configs2         "; bash tools/bv.sh --workload configs2
  echo -n "rep $rep hwq $q c3 b256 defer    "; bash tools/bv.sh --workload configs3 --steps 8 --warmup 3
  echo -n "rep $rep hwq $q c3 b256 seq      "; bash tools/bv.sh --workload configs3 --steps 8 --warmup 3 --group-streams 0
  echo -n "rep $rep hwq $q c3 b64 defer     "; bash tools/bv.sh --workload configs3 --batch 64 --steps 20
 done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_hwq.log
