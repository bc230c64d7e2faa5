#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for k in 0 1 2 3 4 5; do
  echo -n "rep $rep skew $k fp32 "; bash tools/bv.sh --stream-skew $k
  echo -n "rep $rep skew $k bf16 "; bash tools/bv.sh --stream-skew $k --workload configs2
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_stream_skew.log
