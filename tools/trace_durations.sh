#!/bin/bash
# GPU box: every launch duration of kernels matching a pattern, clustered (kernel trace of one bench.py configuration, overlap off)
# usage: tools/trace_durations.sh <name> <sql-like-pattern> <bench args...>  -> gpurun_out/<name>_durations.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; pat=$2; shift; shift
O=$R/gpurun_out/td_$name
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SIMQ_OVERLAP=0 rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --steps 3 --warmup 1 "$@" > $O/kt.out 2> $O/kt.err
cd $R
python - $(find $O -name "kt_results.db") "$pat" > $R/gpurun_out/${name}_durations.txt <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for (kname,) in c.execute("select distinct name from kernels where name like ?", (sys.argv[2],)).fetchall():
    d = sorted(r[0] / 1e3 for r in c.execute("select end-start from kernels where name=?", (kname,)))
    # cluster: new cluster when a duration exceeds the cluster's first by 25 %
    cl, cur = [], [d[0]]
    for v in d[1:]:
        if v > cur[0] * 1.25: cl.append(cur); cur = [v]
        else: cur.append(v)
    cl.append(cur)
    print(kname.replace('simq::(anonymous namespace)::', '')[:70], 'launches', len(d), 'total %.3f ms' % (sum(d) / 1e3))
    for g in cl: print('   n=%3d  %.1f .. %.1f us (mean %.1f)' % (len(g), g[0], g[-1], sum(g) / len(g)))
P
rm -rf $O
cat $R/gpurun_out/${name}_durations.txt
