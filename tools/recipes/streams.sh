#!/bin/bash
# launch-by-launch schedule of one steady-state step (tools/rocprof_streams.py):  tools/lease.sh streams 900 <name> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/st_$name; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 6 --warmup 3 "$@" > $O/kt.out 2> $O/kt.err )
cd $R
DB=$(find $O -name "kt_results.db")
python tools/rocprof_streams.py $DB 2 > gpurun_out/${name}_streams.txt 2> gpurun_out/${name}_streams.err; tail -3 gpurun_out/${name}_streams.err
python tools/rocprof_phases.py $DB > gpurun_out/${name}_phases.txt
rm -rf $O
tail -5 gpurun_out/${name}_phases.txt | cut -c1-300; wc -l gpurun_out/${name}_streams.txt
