#!/bin/bash
# GPU box: per-(kernel, grid) fetch traffic of one workload;  usage: tools/r04_traffic_shapes.sh name [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tshapes; mkdir -p $O
name=$1; shift
B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 5 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o pf_$name -- $B > $O/pf_$name.out 2> $O/pf_$name.err
cd $R
python tools/pmc_traffic_by_shape.py $(find $O -name "pf_${name}_results.db") > $O/fetch_by_shape_${name}.txt 2>&1
if [ -n "$LIST" ]; then python tools/pmc_traffic_by_shape.py $(find $O -name "pf_${name}_results.db") "$LIST" list > $O/fetch_list_${name}.txt 2>&1; fi
find $O -name "*.db" -delete
head -45 $O/fetch_by_shape_${name}.txt
