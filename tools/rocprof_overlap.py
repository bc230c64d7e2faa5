#!/usr/bin/env python3
"""How much of a steady-state window of a rocprofv3 --kernel-trace database has >= 1 / >= 2 / >= 3 kernels running, a matrix kernel running,
ONLY non-matrix kernels running -- for workloads whose steps interleave (several robot groups on launch streams of their own), where
rocprof_phases.py's per-step phases do not exist.  Window = the trace from the `skip`-th clip_sgd_kernel launch on (default: the second
half of them).  Also: per stream / queue busy time, and the summed kernel time per class.
usage: rocprof_overlap.py results.db [fraction of the clip_sgd launches to skip, default 0.5]"""
import collections
import sqlite3
import sys


def klass(n):
    if 'igemm' in n or 'wgrad' in n and 'slab' not in n or 'gemm' in n or 'conv_img' in n or 'stem_conv' in n or 'stem_wgrad_bf16_kernel' in n:
        return 'matrix'
    if 'wino' in n:
        return 'transform'
    if 'bn_' in n or 'chan_reduce' in n or 'stats_fold' in n:
        return 'batchnorm'
    return 'other'


def main():
    c = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sid = 'stream_id' if 'stream_id' in cols else 'queue_id'
    rows = c.execute("select start, end, name, %s from kernels order by start" % sid).fetchall()
    sgd = [r for r in rows if 'clip_sgd_kernel' in r[2]]
    t0 = sgd[int(len(sgd) * skip)][1]
    t1 = sgd[-1][1]
    n_sgd = len([r for r in sgd if t0 < r[1] <= t1])
    win = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in rows if e > t0 and s < t1]
    ev = []
    cls, per_q = collections.Counter(), collections.Counter()
    for s, e, n, q in win:
        m = 1 if klass(n) == 'matrix' else 0
        ev.append((s, 1, m)); ev.append((e, -1, -m))
        cls[klass(n)] += e - s
        per_q[q] += e - s
    ev.sort()
    depth = mdepth = 0
    last = t0
    d_time, m_time, nm_only = collections.Counter(), 0, 0
    for t, dd, dm in ev:
        dt = t - last
        d_time[min(depth, 4)] += dt
        if mdepth > 0:
            m_time += dt
        elif depth > 0:
            nm_only += dt
        depth += dd; mdepth += dm; last = t
    W = t1 - t0
    print('# window %.2f ms = %d optimiser steps (%.3f ms each), %d launches' % (W / 1e6, n_sgd, W / 1e6 / max(n_sgd, 1), len(win)))
    print('kernels running:  0: %.1f %%   1: %.1f %%   2: %.1f %%   3: %.1f %%   >=4: %.1f %%   (>= 2: %.1f %%)' % tuple(
        [100.0 * d_time[k] / W for k in range(5)] + [100.0 * sum(d_time[k] for k in (2, 3, 4)) / W]))
    print('a matrix kernel running: %.1f %% of the window; ONLY non-matrix kernels: %.1f %% (= %.3f ms per optimiser step)' % (
        100.0 * m_time / W, 100.0 * nm_only / W, nm_only / 1e6 / max(n_sgd, 1)))
    print('summed kernel time / window: %.2f   by class (ms per optimiser step): %s' % (
        sum(cls.values()) / W, '  '.join('%s %.3f' % (k, v / 1e6 / max(n_sgd, 1)) for k, v in cls.most_common())))
    print('per stream (busy %% of the window): ' + '  '.join('%s: %.1f' % (q, 100.0 * v / W) for q, v in per_q.most_common()))


if __name__ == '__main__':
    main()
