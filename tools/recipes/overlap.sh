#!/bin/bash
# how many kernels run at once over a steady-state window (tools/rocprof_overlap.py):  tools/lease.sh overlap 900 <name> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/ov_$name; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 6 --warmup 3 "$@" > $O/kt.out 2> $O/kt.err )
cd $R
DB=$(find $O -name "kt_results.db")
python tools/rocprof_overlap.py $DB > gpurun_out/${name}_overlap.txt 2> gpurun_out/${name}_overlap.err
rm -rf $O
cat gpurun_out/${name}_overlap.txt; tail -2 gpurun_out/${name}_overlap.err
