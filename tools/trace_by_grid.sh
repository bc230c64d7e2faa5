#!/bin/bash
# GPU box: kernel trace of one bench.py configuration with the side-stream overlap off, every kernel summarised per launch size
# usage: tools/trace_by_grid.sh <name> <bench args...>  -> gpurun_out/<name>_by_grid.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/tg_$name
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SIMQ_OVERLAP=0 rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --steps 5 --warmup 2 "$@" > $O/kt.out 2> $O/kt.err
cd $R
python - $(find $O -name "kt_results.db") > $R/gpurun_out/${name}_by_grid.txt <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, grid_x/workgroup_x, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name, grid_x order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print('# total %.3f ms over 7 steps = %.3f ms/step' % (tot, tot / 7))
for r in rows[:70]:
    print('%-58s grid %6d calls %4d total %7.3f ms (%4.1f%%) avg %8.2f min %8.2f max %8.2f us' % (r[0].replace('simq::(anonymous namespace)::', '').replace('void ', '').replace('simq::', '')[:58], r[1], r[2], r[3], 100 * r[3] / tot, r[4], r[5], r[6]))
P
rm -rf $O
cat $R/gpurun_out/${name}_by_grid.txt
