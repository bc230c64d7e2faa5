#!/usr/bin/env python3
"""GPU: the fp32 batched transform-domain GEMM (simq_gemm_f32_batched) alone on the shapes of the fp32 step -- us per launch and fraction
of the 157.3 TF/s fp32 matrix peak per (rows, N, K, planes).  usage: tools/gemm_batched_probe.py [batch sizes ...]   (default 32 29)
Ablation build (SIMQ_LIBRARY=.../libsimq_ablate.so): SIMQ_GEMM_NT_RUN etc. apply."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L
st = L.stream_ptr()
PEAK = 157.3e12


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


opts = L.launch_opts(**{k: int(v) for k, v in (kv.split('=') for kv in os.environ.get('PROBE_OPTS', '').split(',') if kv)})
for B in [int(a) for a in sys.argv[1:]] or [32, 29]:
    shapes = []
    for N, K in ((512, 512), (512, 256), (256, 512), (256, 256), (256, 128), (128, 256), (128, 128)):
        shapes.append(('F4 fwd/dgrad', B * 36, N, K, 36))
    for N, K in ((512, 256), (256, 256), (256, 128), (128, 128)):
        shapes.append(('F2 grad fwd', B * 144, N, K, 16))
    if B % 4 == 0:
        for Co, Ci, S in ((512, 512, 1), (512, 256, 2 if B * 36 // 2 >= 512 else 1), (256, 256, 2), (256, 128, 2)):
            shapes.append(('F4 wgrad S=%d' % S, Co, Ci, B * 36 // S, 36 * S))
    tot = 0.0
    for name, M, N, K, P in shapes:
        x = torch.randn(P, M, K, device='cuda'); w = torch.randn(P, N, K, device='cuda'); y = torch.empty(P, M, N, device='cuda')
        us = timeit(lambda: L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y), M, N, K, P, st, opts=opts))
        fl = 2.0 * M * N * K * P
        tot += us
        print('B=%3d %-14s M=%5d N=%4d K=%5d x%3d  %7.1f us  %6.1f TF/s  %.3f of peak   (%.1f MB operands)' % (
            B, name, M, N, K, P, us, fl / us / 1e6, fl / us / 1e6 / (PEAK / 1e12), 4e-6 * P * (M * K + N * K + M * N)), flush=True)
    print('B=%3d sum %.1f us' % (B, tot))
