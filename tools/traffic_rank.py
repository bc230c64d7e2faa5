#!/usr/bin/env python3
"""Rank the kernels of a step by HBM bytes moved, from the per-kernel PMC summary the `traffic` recipe leaves
(profiles/rNN_pmc_traffic*.json: {kernel: {launches, bytes_per_launch, read_bytes_per_launch, write_bytes_per_launch}}).
Steps are counted by the clip_sgd_kernel launches (one per train() call).

    python tools/traffic_rank.py profiles/r05_pmc_traffic.json [--top 25]
"""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('summary')
    ap.add_argument('--top', type=int, default=25)
    args = ap.parse_args()
    d = json.load(open(args.summary))
    steps = d['clip_sgd_kernel']['launches']
    rows = sorted(((v['bytes_per_launch'] * v['launches'] / steps, v['launches'] / steps, v['bytes_per_launch'],
                    v['read_bytes_per_launch'], v['write_bytes_per_launch'], k) for k, v in d.items()), reverse=True)
    total = sum(r[0] for r in rows)
    print('# %s: %d steps, %.2f GB of HBM traffic per step over %d kernels' % (args.summary, steps, total / 1e9, len(rows)))
    print('%10s %6s %8s %10s %8s %8s  %s' % ('MB/step', '%', 'n/step', 'MB/launch', 'read', 'write', 'kernel'))
    for b, n, bl, rl, wl, k in rows[:args.top]:
        print('%10.1f %6.1f %8.1f %10.1f %8.1f %8.1f  %s' % (b / 1e6, 100 * b / total, n, bl / 1e6, rl / 1e6, wl / 1e6, k[:100]))
    rest = rows[args.top:]
    if rest:
        print('%10.1f %6.1f %8s %10s %8s %8s  (%d more kernels)' % (sum(r[0] for r in rest) / 1e6, 100 * sum(r[0] for r in rest) / total, '', '', '', '', len(rest)))


if __name__ == '__main__':
    main()
