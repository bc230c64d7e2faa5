#!/bin/bash
# F(4x4,3x3) transform kernels per shape, alone on the device (rocprofv3 --kernel-trace over tools/wino4f_probe.py):  tools/lease.sh w4f 600
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=$R/gpurun_out/w4f; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o kt -- python $R/tools/wino4f_probe.py 32 29 > $O/kt.out 2> $O/kt.err )
DB=$(find $O -name "kt_results.db")
echo "$(grep -c 'max err' $O/kt.out) shapes ok"; grep -i "assert\|error" $O/kt.out $O/kt.err | head -3
python tools/wino4f_trace.py $DB 32 29 | tee gpurun_out/w4f.log
rm -rf $O
