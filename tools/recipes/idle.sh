#!/bin/bash
# GPU box: how much of a step is the device idle?  Kernel trace of one bench.py configuration (streams as in production); per step the
# wall span from the first kernel's start to the last kernel's end against the UNION of the kernels' busy intervals.
# usage: tools/lease.sh idle 600 <name> <bench args...>  -> gpurun_out/<name>_idle.txt
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
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
rows4 = rows
rows = [r[:3] for r in rows]
# steps are delimited by clip_sgd_kernel (last kernel of a train step)
ends = [i for i, r in enumerate(rows) if 'clip_sgd' in r[0]]
print('# %d kernels, %d steps' % (len(rows), len(ends)))
for a, b in zip(ends[:-1], ends[1:]):
    ks = rows[a + 1:b + 1]
    span = ks[-1][2] - ks[0][1]
    busy = 0; cur_s, cur_e = ks[0][1], ks[0][2]
    gaps = []
    big = []
    last_name = ks[0][0]
    for nm, s, e in ks[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append(s - cur_e)
            if s - cur_e > 20000: big.append('%.0f us between %s and %s' % ((s - cur_e) / 1e3, last_name.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][-40:], nm.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][-40:]))
            cur_s, cur_e = s, e
            last_name = nm
        else:
            if e > cur_e: last_name = nm
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for _, s, e in ks)
    gaps.sort()
    print('step: %4d kernels  span %.3f ms  busy(union) %.3f ms  idle %.3f ms (%.1f%%)  kernel sum %.3f ms  gaps: n=%d median %.2f us, >5us: %d (%.3f ms), >20us: %d (%.3f ms)' % (
        len(ks), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, ksum / 1e6, len(gaps),
        gaps[len(gaps) // 2] / 1e3 if gaps else 0, sum(g > 5000 for g in gaps), sum(g for g in gaps if g > 5000) / 1e6,
        sum(g > 20000 for g in gaps), sum(g for g in gaps if g > 20000) / 1e6))
    for t in big: print('    gap: ' + t)
    if a == ends[-2]:      # the neighbourhood of the last step's largest gap: (offset from the gap's start in us, duration, stream, kernel)
        best, cur = (0, 0), ks[0][2]
        for i in range(1, len(ks)):
            if ks[i][1] - cur > best[0]: best = (ks[i][1] - cur, i)
            cur = max(cur, ks[i][2])
        i0 = a + 1 + best[1]
        t0 = rows4[i0][1]
        for r in rows4[max(i0 - 8, 0):i0 + 8]:
            print('      %+9.1f us  %7.1f us  stream %s  %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][-60:]))
P
rm -rf $O
cat $R/gpurun_out/${name}_idle.txt
