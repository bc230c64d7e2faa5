#!/bin/bash
# GPU box: kernel-trace timelines (busy / overlapped / idle fractions, gap attribution) of both bench workloads
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
for w in b32: bf16_b128:"--workload configs2"; do
  name=${w%%:*}; extra=${w#*:}
  B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 10 --warmup 3 $extra"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt_$name -- $B > $O/kt_$name.out 2> $O/kt_$name.err)
  python $R/tools/rocprof_timeline.py $(find $O -name "kt_${name}_results.db") 0.5 > $O/timeline_$name.txt
  cat $O/timeline_$name.txt
done
find $O -name "*.db" -delete
