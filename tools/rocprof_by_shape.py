#!/usr/bin/env python3
"""Per-(kernel, grid) durations of a rocprofv3 --kernel-trace database: one line per launch shape, steady-state launches only.
usage: rocprof_by_shape.py results.db steps [skip_fraction=0.5]   (steps = bench steps inside the kept fraction, for the per-step column)"""
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], float(sys.argv[2])
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    c = sqlite3.connect(db)
    rows = c.execute("select start, end, name, grid_x / workgroup_x, grid_y, grid_z from kernels order by start").fetchall()
    rows = rows[int(len(rows) * skip):]
    agg = {}
    for s, e, name, gx, gy, gz in rows:
        name = name.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:56]
        k = (name, gx, gy)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('# %d launches kept, %.3f ms of kernel time, %.3f ms per step' % (len(rows), tot / 1e3, tot / 1e3 / steps))
    print('%-56s %7s %5s %9s %9s %9s %9s %9s' % ('kernel', 'blocks', 'gy', 'n/step', 'avg_us', 'min_us', 'max_us', 'us/step'))
    for (name, gx, gy), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print('%-56s %7d %5d %9.2f %9.2f %9.2f %9.2f %9.1f' % (name, gx, gy, a[0] / steps, a[1] / a[0], a[2], a[3], a[1] / steps))


if __name__ == '__main__':
    main()
