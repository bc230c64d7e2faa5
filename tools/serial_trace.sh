#!/bin/bash
# GPU box: kernel trace of the fp32 step with the side-stream overlap off (every kernel alone on the GPU), summarised per
# kernel and, for the elementwise kernels, per launch size.   usage: tools/serial_trace.sh [extra bench flags]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/serial
rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --steps 5 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
SIMQ_OVERLAP=0 rocprofv3 --kernel-trace -d $O -o kt -- $B > $O/kt.out 2> $O/kt.err
cd $R
DB=$(find $O -name "kt_results.db")
python tools/rocprof_summary.py $DB 7 > $O/kernel_trace.txt
python - $DB > $O/by_grid.txt <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for pat in ('bn_apply', 'bn_bwd_apply', 'upsample2x_fwd', 'chan_reduce', 'igemm_conv_kernel<64, 64, true, false>', 'igemm_conv_kernel<96, 32', 'igemm_conv_kernel<32, 32', 'igemm_conv_kernel<96, 64', 'igemm_conv_kernel<64, 32', 'igemm_conv_kernel<64, 64, false', 'wgrad_kernel', 'wino'):
    rows = c.execute("select name, grid_x/workgroup_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like ? group by name, grid_x order by 1, 2", ('%' + pat + '%',)).fetchall()
    for r in rows:
        print('%-60s grid %6d  calls %4d  avg %8.2f us  min %8.2f us' % (r[0].replace('simq::(anonymous namespace)::', '').replace('void ', '')[:60], r[1], r[2], r[3], r[4]))
P
find $O -name "*.db" -size +30M -delete
head -30 $O/kernel_trace.txt | cut -c1-150
cat $O/by_grid.txt
