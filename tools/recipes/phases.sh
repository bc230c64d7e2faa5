#!/bin/bash
# kernel-trace phase breakdown of both bench legs (tools/rocprof_phases.py) -> gpurun_out/phases_*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ph; rm -rf $O; mkdir -p $O
for w in b32: bf16_b128:"--workload configs2"; do
  name=${w%%:*}; extra=${w#*:}
  B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 10 --warmup 3 $extra $*"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt_$name -- $B > $O/kt_$name.out 2> $O/kt_$name.err)
  python $R/tools/rocprof_phases.py $(find $O -name "kt_${name}_results.db") > $R/gpurun_out/phases_$name.txt
  cat $R/gpurun_out/phases_$name.txt
done
find $O -name "*.db" -delete
