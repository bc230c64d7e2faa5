#!/bin/bash
# GPU box: rocprofv3 kernel trace of one bench.py configuration -> gpurun_out/<name>_kernel_trace.txt
# usage: tools/kt.sh <name> <bench.py args...>      (7 full steps, nothing else: --no-cpu-baseline --no-extras --no-m1 --no-roofline)
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/kt_$name
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 5 --warmup 2 "$@" > $O/kt.out 2> $O/kt.err
cd $R
python tools/rocprof_summary.py $(find $O -name "kt_results.db") 7 > $R/gpurun_out/${name}_kernel_trace.txt
rm -rf $O
head -45 $R/gpurun_out/${name}_kernel_trace.txt
