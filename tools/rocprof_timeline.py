#!/usr/bin/env python3
"""GPU timeline of a rocprofv3 --kernel-trace database: how much of the wall time of the steady-state steps the device runs at least one
kernel, how much it runs two (the side stream's overlap), how much it idles, and which kernels the idle gaps sit in front of.
usage: rocprof_timeline.py results.db [skip_fraction=0.4]   (the first skip_fraction of the launches -- set-up and warm-up -- is dropped)
Note: the tracer slows the host's launch path; gaps seen here are an upper bound of those of an untraced run."""
import collections
import sqlite3
import sys


def main():
    db = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    c = sqlite3.connect(db)
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy1 = busy2 = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    span = t1 - t0
    print('# timeline of %d launches, %.3f ms: >= 1 kernel running %.1f %%, >= 2 running %.1f %%, idle %.1f %% (%.3f ms)' % (
        len(rows), span / 1e6, 100.0 * busy1 / span, 100.0 * busy2 / span, 100.0 * (span - busy1) / span, (span - busy1) / 1e6))
    # idle gaps: time with nothing running, attributed to the kernel that ends the gap
    gaps = collections.defaultdict(lambda: [0, 0.0])
    hist = collections.Counter()
    cur_end = rows[0][1]
    for s, e, name in rows[1:]:
        if s > cur_end:
            g = (s - cur_end) / 1e3
            name = name.replace('simq::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
            gaps[name][0] += 1
            gaps[name][1] += g
            hist['<1us' if g < 1 else '1-2us' if g < 2 else '2-5us' if g < 5 else '5-20us' if g < 20 else '>=20us'] += 1
        cur_end = max(cur_end, e)
    print('# idle gaps by length:', dict(hist))
    print('%-62s %7s %10s %8s' % ('idle gap in front of', 'gaps', 'total_us', 'avg_us'))
    for name, (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%-62s %7d %10.1f %8.2f' % (name, n, tot, tot / n))


if __name__ == '__main__':
    main()
