#!/bin/bash
# kernel trace of one bench.py configuration, per kernel and per (kernel, grid):  tools/lease.sh kt 900 <name> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/kt_$name; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 5 --warmup 2 "$@" > $O/kt.out 2> $O/kt.err )
cd $R
DB=$(find $O -name "kt_results.db")
python tools/rocprof_summary.py $DB 7 > gpurun_out/${name}_kernel_trace.txt
python tools/rocprof_by_shape.py $DB 3.5 0.5 > gpurun_out/${name}_launch_shapes.txt
python tools/rocprof_phases.py $DB > gpurun_out/${name}_phases.txt
rm -rf $O
head -45 gpurun_out/${name}_kernel_trace.txt | cut -c1-170; cat gpurun_out/${name}_phases.txt
