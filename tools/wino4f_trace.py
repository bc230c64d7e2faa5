#!/usr/bin/env python3
"""Reads the rocprofv3 --kernel-trace database of a tools/wino4f_probe.py run: per shape (12 operator calls each, in the probe's order) the
median duration of the input transform, the batched GEMM and the output transform, with the transforms' HBM rate.
usage: wino4f_trace.py results.db B [B ...]   (the probe's batch sizes, same order)"""
import sqlite3
import statistics
import sys

SHAPES = [(512, 512), (256, 512), (512, 256), (256, 256), (128, 256), (256, 128), (128, 128)]
db, batches = sys.argv[1], [int(a) for a in sys.argv[2:]]
rows = sqlite3.connect(db).execute("select start, end, name, grid_x / workgroup_x from kernels order by start").fetchall()
calls, cur = [], None
for s, e, name, gx in rows:
    d = (e - s) / 1e3
    if 'wino_weight_all_kernel' in name:
        cur = {}
        calls.append(cur)
    elif cur is not None:
        for key in ('wino4f_input', 'igemm_conv_kernel', 'wino4f_output'):
            if key in name and key not in cur:
                cur[key] = (d, gx, name.split('wino4f_')[-1].split('(')[0] if 'wino4f' in name else '')
calls = [c for c in calls if len(c) == 3]
assert len(calls) == 12 * len(SHAPES) * len(batches), len(calls)
tot_in = tot_out = 0.0
for i, B in enumerate(batches):
    for j, (cin, cout) in enumerate(SHAPES):
        grp = calls[(i * len(SHAPES) + j) * 12 + 2:(i * len(SHAPES) + j + 1) * 12]
        t_in = statistics.median(c['wino4f_input'][0] for c in grp)
        t_g = statistics.median(c['igemm_conv_kernel'][0] for c in grp)
        t_out = statistics.median(c['wino4f_output'][0] for c in grp)
        px = B * 576 * 4.0
        mb_in, mb_out = px * cin * 3.25 / 1e6, px * cout * 3.25 / 1e6       # x + 2.25 x (V)  /  2.25 y (Mt) + y
        tot_in += t_in; tot_out += t_out
        print('B=%2d %3d->%3d  input %-22s %4d blocks %6.1f us %4.1f TB/s | gemm %6.1f us | output %-22s %4d blocks %6.1f us %4.1f TB/s'
              % (B, cin, cout, grp[0]['wino4f_input'][2], grp[0]['wino4f_input'][1], t_in, mb_in / t_in, t_g,
                 grp[0]['wino4f_output'][2], grp[0]['wino4f_output'][1], t_out, mb_out / t_out))
print('sum of medians: input %.1f us, output %.1f us' % (tot_in, tot_out))
