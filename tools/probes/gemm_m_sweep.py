#!/usr/bin/env python3
"""GPU: does the exact-fp32 batched GEMM lose time to the power-of-two plane strides of B = 32?  simq_gemm_f32_batched on layer4's
F(4x4) problem (N = K = 512, 36 planes) and on the F(2x2) grad-forward problem (N = 512, K = 256, 16 planes) over row counts around the
bench's (1152 = 32 x 36, 4608 = 32 x 144): equal tile counts with a plane stride that is / is not a multiple of 256 KB separate block
quantisation from channel aliasing.  usage: tools/probes/gemm_m_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L
st = L.stream_ptr()
PEAK = 157.3


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def warm_clocks(seconds=2.0):
    """The first second of work on an idle device runs at ramping clocks (the first shapes of an unwarmed sweep read 15-20 % slow)."""
    import time
    x = torch.randn(36, 1152, 512, device='cuda'); w = torch.randn(36, 512, 512, device='cuda'); y = torch.empty(36, 1152, 512, device='cuda')
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y), 1152, 512, 512, 36, st)
        torch.cuda.synchronize()


warm_clocks()


for N, K, P, Ms in ((512, 512, 36, (1044, 1088, 1092, 1100, 1144, 1148, 1152, 1156, 1160, 1216)),
                    (512, 256, 16, (4176, 4544, 4600, 4604, 4608, 4612, 4672)),
                    (256, 256, 36, (1044, 1100, 1148, 1152, 1156)),
                    (512, 1152, 36, ()),):
    for M in Ms:
        x = torch.randn(P, M, K, device='cuda'); w = torch.randn(P, N, K, device='cuda'); y = torch.empty(P, M, N, device='cuda')
        us = timeit(lambda: L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y), M, N, K, P, st))
        tiles = (M + 63) // 64 * (N // 64) * P
        fl = 2.0 * M * N * K * P
        print('N=%4d K=%4d x%2d  M=%5d (%5d tiles, x plane stride %% 256 KB = %6d)  %7.1f us  %6.1f TF/s  %.3f of peak   %.4f us/tile' % (
            N, K, P, M, tiles, (M * K * 4) % (256 * 1024), us, fl / us / 1e6, fl / us / 1e6 / PEAK, us / tiles), flush=True)
