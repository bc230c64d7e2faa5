#!/usr/bin/env python3
"""FETCH_SIZE (read side, gfx950 correction as in pmc_traffic.py) per launch and per (kernel, grid) from one rocprofv3 --pmc FETCH_SIZE
database -- the per-layer view of the per-kernel averages of pmc_traffic.py.
usage: pmc_traffic_by_shape.py fetch.db [name-substring [list]]     (list: every matching dispatch of the last 15 % of the run, in order,
with its duration under the counter pass and the read rate that implies)"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    if 'grid_x' in cols:
        q = ("select name, grid_x / workgroup_x, grid_y, dispatch_id, sum(counter_value) from pmc_events where counter_name='FETCH_SIZE' "
             "group by dispatch_id")
    else:
        print('# pmc_events columns:', cols)
        print('# kernels columns:', kcols)
        q = ("select p.name, k.grid_x / k.workgroup_x, k.grid_y, p.dispatch_id, sum(p.counter_value) from pmc_events p join kernels k "
             "on k.dispatch_id = p.dispatch_id where p.counter_name='FETCH_SIZE' group by p.dispatch_id")
    if len(sys.argv) > 3 and sys.argv[3] == 'list':
        q2 = ("select p.name, k.grid_x / k.workgroup_x, p.dispatch_id, sum(p.counter_value), k.end - k.start, k.start from pmc_events p join kernels k "
              "on k.dispatch_id = p.dispatch_id where p.counter_name='FETCH_SIZE' group by p.dispatch_id order by k.start")
        rows = [r for r in c.execute(q2)]
        rows = rows[int(len(rows) * 0.85):]
        for name, gx, did, v, dur, _ in rows:
            name = name.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:56]
            if pat in name:
                mb = 2.0 * 1024.0 * v / 1e6
                print('%-56s %6d blocks  fetch %8.1f MB  %8.1f us  read %5.2f TB/s' % (name, gx, mb, dur / 1e3, mb / (dur / 1e3) * 1e-6 * 1e6 / 1e6 if dur else 0))
        return
    agg = {}
    for name, gx, gy, _, v in c.execute(q):
        name = name.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:56]
        if pat not in name:
            continue
        a = agg.setdefault((name, gx, gy), [0, 0.0, 1e30, 0.0])
        mb = 2.0 * 1024.0 * v / 1e6
        a[0] += 1; a[1] += mb; a[2] = min(a[2], mb); a[3] = max(a[3], mb)
    print('%-56s %7s %5s %7s %10s %10s %10s' % ('kernel', 'blocks', 'gy', 'n', 'avg_MB', 'min_MB', 'max_MB'))
    for (name, gx, gy), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print('%-56s %7d %5d %7d %10.1f %10.1f %10.1f' % (name, gx, gy, a[0], a[1] / a[0], a[2], a[3]))


if __name__ == '__main__':
    main()
