#!/bin/bash
# GPU box: how much of a step is the device idle?  Kernel trace of one bench.py configuration (streams as in production); per step the
# wall span from the first kernel's start to the last kernel's end against the UNION of the kernels' busy intervals.
# usage: tools/idle_gaps.sh <name> <bench args...>  -> gpurun_out/<name>_idle.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/ig_$name
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --steps 6 --warmup 2 "$@" > $O/kt.out 2> $O/kt.err
cd $R
python - $(find $O -name "kt_results.db") > $R/gpurun_out/${name}_idle.txt <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# steps are delimited by clip_sgd_kernel (last kernel of a train step)
ends = [i for i, r in enumerate(rows) if 'clip_sgd' in r[0]]
print('# %d kernels, %d steps' % (len(rows), len(ends)))
for a, b in zip(ends[:-1], ends[1:]):
    ks = rows[a + 1:b + 1]
    span = ks[-1][2] - ks[0][1]
    busy = 0; cur_s, cur_e = ks[0][1], ks[0][2]
    gaps = []
    for _, s, e in ks[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for _, s, e in ks)
    gaps.sort()
    print('step: %4d kernels  span %.3f ms  busy(union) %.3f ms  idle %.3f ms (%.1f%%)  kernel sum %.3f ms  gaps: n=%d median %.2f us, >5us: %d (%.3f ms), >20us: %d (%.3f ms)' % (
        len(ks), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, ksum / 1e6, len(gaps),
        gaps[len(gaps) // 2] / 1e3 if gaps else 0, sum(g > 5000 for g in gaps), sum(g for g in gaps if g > 5000) / 1e6,
        sum(g > 20000 for g in gaps), sum(g for g in gaps if g > 20000) / 1e6))
P
rm -rf $O
cat $R/gpurun_out/${name}_idle.txt
