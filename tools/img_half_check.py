#!/usr/bin/env python
"""Per-launch time of the bf16 image-tile kernel's HALF-map form (12 image rows x 128 channels per block, conv_igemm_bf16_img.hip) against
the whole-map form and the 288 x 128 LDS-DMA tile, alone on the device (GPU box).  Each call converts x and w to bf16 planes first (the
op-level entry point): the same constant on every row; it is measured separately and subtracted."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
from simq import _lib as L  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bench(B, Cin, Cout):
    H, k = 24, 3
    x = torch.randn(B, H, H, Cin, device='cuda')
    w = torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * 9) ** 0.5
    y = torch.empty(B, H, H, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    st = L.stream_ptr()
    gf = 2.0 * B * 576 * Cout * 9 * Cin / 1e9
    # the conversion constant: a 1x1 problem of the same x costs the two plane conversions + a negligible GEMM? no -- time the split alone
    row = []
    for name, tile in (('auto', (0, 0)), ('half-map 1288x128', (1288, 128)), ('whole-map 576x128', (576, 128)), ('dma 288x128', (288, 128)), ('reg 96x128', (96, 128))):
        L.lib.call('simq_tune_force_tile', *tile)
        try:
            us = timeit(lambda: L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, 1, L.ptr(scratch), L.ptr(stats), st))
        finally:
            L.lib.call('simq_tune_force_tile', 0, 0)
        row.append('%s %6.1f us' % (name, us))
    print('B=%3d %3d->%3d (%.1f GFLOP): ' % (B, Cin, Cout, gf) + ' | '.join(row), flush=True)


for B, Cin, Cout in ((128, 128, 128), (128, 64, 128), (115, 128, 128), (64, 256, 256), (64, 512, 512), (64, 256, 512), (128, 256, 256)):
    bench(B, Cin, Cout)
