#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace [--pmc ...]) as text: per-kernel launch count,
total / avg / min / max duration, register and LDS use, and PMC counter sums when present.
usage: rocprof_summary.py results.db [steps]   (steps: divide totals to report per-step time)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x/workgroup_x) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print('# rocprofv3 kernel-trace summary of %s' % db)
    print('# total kernel time %.3f ms over %d launches%s' % (tot, sum(r[1] for r in rows),
          '' if steps is None else ' ; %.3f ms per step (%g steps incl. warm-up)' % (tot / steps, steps)))
    print('%-72s %6s %10s %6s %9s %9s %9s %5s %5s %5s %7s %7s' % ('kernel', 'calls', 'total_ms', '%', 'avg_us', 'min_us', 'max_us', 'vgpr', 'agpr', 'sgpr', 'lds_B', 'maxgrid'))
    for r in rows:
        name = r[0].replace('simq::(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0] if len(name) > 72 else name
        print('%-72s %6d %10.3f %6.1f %9.2f %9.2f %9.2f %5d %5d %5d %7d %7d' % (name[:72], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5], r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
    try:
        pm = c.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                       "group by name, counter_name order by name").fetchall()
    except sqlite3.Error as e:
        pm = []
        print('# (no PMC data: %s)' % e)
    if pm:
        print('\n# PMC counters (sum over launches; per-launch = sum / calls)')
        for name, ctr, n, v in pm:
            name = name.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            print('%-72s %-28s calls=%6d sum=%.6g per_launch=%.6g' % (name[:72], ctr, n, v, v / max(n, 1)))


if __name__ == '__main__':
    main()
