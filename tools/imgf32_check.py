#!/usr/bin/env python
"""Per-launch time of the fp32 image-tile kernel (conv_img_f32.hip) against the implicit-GEMM tiles it replaces: the 3x3 convolutions
with 64 input channels on the 24x24 maps, alone on the device (GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
from simq import _lib as L  # noqa: E402


def bench(B, Cout, tile, stats_on, reps=50):
    Cin, H, k = 64, 24, 3
    x = torch.randn(B, H, H, Cin, device='cuda')
    w = torch.randn(Cout, k, k, Cin, device='cuda') / 24.0
    y = torch.empty(B, H, H, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda') if stats_on else None
    st = L.stream_ptr()
    opts = L.launch_opts(tile=tile)
    try:
        def go():
            L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, L.ptr(stats), st, opts=opts)
        for _ in range(5):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            go()
        e1.record()
        torch.cuda.synchronize()
    finally:
        pass
    return e0.elapsed_time(e1) / reps * 1e3


for B in (32, 29, 128):
    for Cout in (64, 128):
        gf = 2.0 * B * 576 * Cout * 576 / 1e9
        row = []
        for tile in ((0, 0), (32, 32), (64, 32), (64, 64), (96, 64)):
            us = bench(B, Cout, tile, True)
            row.append('%s %6.1f us (%5.1f TF/s)' % ('img' if tile == (0, 0) else '%dx%d' % tile, us, gf / us * 1e3))
        print('B=%3d 64->%3d  %.2f GFLOP: ' % (B, Cout, gf) + ' | '.join(row))
