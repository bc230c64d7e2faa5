#!/bin/bash
# GPU box: per-(kernel, grid) launch durations of both bench workloads (steady state: the second half of a 3 + 10 step run ~ 6.5 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/shapes; rm -rf $O; mkdir -p $O
for w in b32: bf16_b128:"--workload configs2"; do
  name=${w%%:*}; extra=${w#*:}
  B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 10 --warmup 3 $extra"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt_$name -- $B > $O/kt_$name.out 2> $O/kt_$name.err)
  python $R/tools/rocprof_by_shape.py $(find $O -name "kt_${name}_results.db") 6.5 0.5 > $O/shapes_$name.txt
done
find $O -name "*.db" -delete
