#!/bin/bash
# alternating A/B of ablation-build switches on ONE box (tools/ab_step.py):  tools/lease.sh abenv 900 REPS WORKLOAD STEPS "VAR=a" "VAR=b" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
reps=$1; wl=$2; steps=$3; shift 3
for rep in $(seq 1 $reps); do
  for spec in "$@"; do echo -n "rep $rep: "; env $spec python tools/ab_step.py $wl $steps 2>&1 | tail -1; done
done 2>&1 | tee gpurun_out/abenv.log
