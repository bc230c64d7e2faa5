#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of one workload (HBM-side bytes per kernel and per step):  tools/lease.sh traffic 900 <name> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/traffic; mkdir -p $O
name=$1; shift
B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 5 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o pf_$name -- $B > $O/pf_$name.out 2> $O/pf_$name.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o pw_$name -- $B > $O/pw_$name.out 2> $O/pw_$name.err
cd $R
python tools/pmc_traffic.py $(find $O -name "pf_${name}_results.db") $(find $O -name "pw_${name}_results.db") $O/pmc_traffic_${name}.json > $O/bench_${name}_pmc_traffic.txt
python tools/pmc_traffic_by_shape.py $(find $O -name "pf_${name}_results.db") > $O/fetch_by_shape_${name}.txt 2>&1
find $O -name "*.db" -delete
head -14 $O/bench_${name}_pmc_traffic.txt
