#!/bin/bash
# alternating A/B of bench.py settings on ONE box (boxes differ by ~2 %): tools/lease.sh ab 900 REPS "label:bench args" "label:bench args" ...
#   each run: tools/bv.sh (30 timed steps + 5 warm-up, no CPU baseline / extras / roofline) -> M2 / M1 / ms per step
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
reps=$1; shift
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    label=${spec%%:*}; args=${spec#*:}
    echo -n "rep $rep  $label  "; bash tools/bv.sh $args
  done
done 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/ab.log
