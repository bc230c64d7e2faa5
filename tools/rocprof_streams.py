#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace database launch by launch: start (us from the step's first kernel), duration, stream,
kernel -- the schedule the streams of simq_train_step produced.  A step = the kernels between two clip_sgd_kernel launches.
usage: rocprof_streams.py results.db [step index from the end, default 2] [min duration us to list, default 0]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sid = 'stream_id' if 'stream_id' in cols else 'queue_id'
    rows = c.execute("select start, end, name, %s from kernels order by start" % sid).fetchall()
    sgd = [i for i, r in enumerate(rows) if 'clip_sgd_kernel' in r[2]]
    a, b = sgd[-back - 1], sgd[-back]
    step = rows[a + 1:b + 1]
    t0 = step[0][0]
    streams = sorted({r[3] for r in step}, key=lambda s: -sum(r[1] - r[0] for r in step if r[3] == s))
    label = {s: chr(ord('A') + i) for i, s in enumerate(streams)}
    print('# step of %.1f us, %d launches; streams by busy time: %s' % ((step[-1][1] - t0) / 1e3, len(step), '  '.join(
        '%s=%s (%.0f us)' % (label[s], s, sum(r[1] - r[0] for r in step if r[3] == s) / 1e3) for s in streams)))
    last_end = {}
    for s, e, name, st in step:
        gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
        last_end[st] = e
        if (e - s) / 1e3 < min_us:
            continue
        print('%9.1f %8.1f  %s%s  %s%s' % ((s - t0) / 1e3, (e - s) / 1e3, '  ' * (ord(label[st]) - 65), label[st], name.replace('(anonymous namespace)::', '').replace('simq::', '').replace('void ', '').split('(')[0][:70],
                                        '   [stream idle %.0f us before]' % gap if gap > 15 else ''))


if __name__ == '__main__':
    main()
