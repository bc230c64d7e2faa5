#!/usr/bin/env python3
"""Phases of the TD step from a rocprofv3 --kernel-trace database of bench.py: per steady-state step the wall time of the forward phase
(three forwards, up to td_huber), the backward phase (up to the gradient-norm kernel) and the tail (clip + SGD + weight cache), and inside
each phase the summed kernel time per class and how much of the phase has >= 1 / >= 2 kernels running.
usage: rocprof_phases.py results.db"""
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
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    print('# kernels columns:', cols)
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
    short = lambda n: n.replace('simq::(anonymous namespace)::', '').replace('simq::', '').replace('void ', '').split('(')[0]
    rows = [(s, e, short(n)) for s, e, n in rows]
    td = [i for i, r in enumerate(rows) if r[2].startswith('td_huber')]
    sq = [i for i, r in enumerate(rows) if r[2].startswith('sumsq')]
    wp = [i for i, r in enumerate(rows) if r[2].startswith('weight_prep_all') or r[2].startswith('wino_weight_all')]
    steps = []
    for k in range(1, len(td)):
        t_td = rows[td[k]][0]
        prev_end = max(rows[i][1] for i in wp if rows[i][0] < t_td)            # the previous step's last weight-cache kernel
        nsq = [i for i in sq if rows[i][0] > t_td]
        if not nsq:
            break
        t_sq = rows[nsq[0]][0]
        nwp = [rows[i][1] for i in wp if rows[i][0] > t_sq]
        if not nwp:
            break
        nxt_td = rows[td[k + 1]][0] if k + 1 < len(td) else None
        t_end = max(e for e in nwp if nxt_td is None or e < nxt_td)
        steps.append((prev_end, t_td, t_sq, t_end))
    steps = steps[len(steps) // 2:]                                              # steady state: the second half
    agg = collections.OrderedDict((ph, dict(wall=0.0, b1=0.0, b2=0.0, mat=0.0, cls=collections.Counter(), lonely=collections.Counter())) for ph in ('forward', 'backward', 'tail'))
    for (a, b, cc, d) in steps:
        for ph, lo, hi in (('forward', a, b), ('backward', b, cc), ('tail', cc, d)):
            ev = []
            for s, e, n in rows:
                if e <= lo or s >= hi:
                    continue
                s2, e2 = max(s, lo), min(e, hi)
                m = 1 if klass(n) == 'matrix' else 0
                ev.append((s2, 1, m, n)); ev.append((e2, -1, -m, n))
                agg[ph]['cls'][klass(n)] += (e2 - s2) / 1e3
            ev.sort(key=lambda t: (t[0], t[1]))
            depth, mdepth, last = 0, 0, lo
            running = collections.Counter()
            for t, dd, dm, n in ev:
                if depth >= 1:
                    agg[ph]['b1'] += (t - last) / 1e3
                if depth >= 2:
                    agg[ph]['b2'] += (t - last) / 1e3
                if mdepth >= 1:
                    agg[ph]['mat'] += (t - last) / 1e3
                elif depth >= 1:                      # only non-matrix kernels on the device: attribute the interval to them
                    for k, c in running.items():
                        if c > 0:
                            agg[ph]['lonely'][k] += (t - last) / 1e3 / sum(1 for v in running.values() if v > 0)
                depth += dd
                mdepth += dm
                running[n] += dd
                last = t
            agg[ph]['wall'] += (hi - lo) / 1e3
    n = max(len(steps), 1)
    print('# %d steady-state steps; per step, microseconds' % len(steps))
    print('%-9s %9s %9s %9s %11s   %s' % ('phase', 'wall', '>=1 busy', '>=2 busy', 'matrix busy', 'summed kernel time by class'))
    for ph, v in agg.items():
        print('%-9s %9.1f %9.1f %9.1f %11.1f   %s' % (ph, v['wall'] / n, v['b1'] / n, v['b2'] / n, v['mat'] / n, '  '.join('%s %.1f' % (k, t / n) for k, t in v['cls'].most_common())))
    print('# wall time with NO matrix-class kernel on the device, by the kernels that ran then (us per step):')
    for ph, v in agg.items():
        print('%-9s %s' % (ph, '  '.join('%s %.1f' % (k[:34], t / n) for k, t in v['lonely'].most_common(12))))
    print('step wall %.1f us' % (sum(v['wall'] for v in agg.values()) / n))


if __name__ == '__main__':
    main()
