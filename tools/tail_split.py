#!/usr/bin/env python3
"""GPU tuning aid: the fp32 implicit-GEMM kernel with and without the balanced last round (K-sliced tail tiles) on the
layer shapes / batch sizes of the headline step; checks that both produce the same output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L

SHAPES = [('l4', 512, 512, 3), ('l4a', 256, 512, 3), ('dg_l4a', 512, 256, 3), ('l3', 256, 256, 3), ('l2', 128, 128, 3), ('l1', 64, 64, 3), ('ds4', 256, 512, 1)]
st = L.stream_ptr()


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B in [int(a) for a in sys.argv[1:]] or [32, 29, 27, 64]:
    for name, Cin, Cout, k in SHAPES:
        H, pad = 24, k // 2
        x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
        ys, ms = [], []
        for on in (0, 1):
            L.lib.call('simq_tune_tail_split', on)
            y = torch.empty(B, H, H, Cout, device='cuda')
            f = lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, pad, None, st)
            ms.append(timeit(f)); ys.append(y)
        flops = 2.0 * B * H * H * Cout * k * k * Cin
        err = float((ys[0] - ys[1]).abs().max() / ys[0].abs().max())
        print('B=%3d %-7s off %6.1f TF (%.3f ms)  on %6.1f TF (%.3f ms)  %+5.1f %%   max rel diff %.1e'
              % (B, name, flops / ms[0] / 1e9, ms[0], flops / ms[1] / 1e9, ms[1], 100 * (ms[0] / ms[1] - 1), err))

print('--- forced tiles, balanced last round on ---')
L.lib.call('simq_tune_tail_split', 1)
for B in (32, 29):
    for name, Cin, Cout, k in SHAPES[:4]:
        H, pad = 24, k // 2
        x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
        y = torch.empty(B, H, H, Cout, device='cuda')
        flops = 2.0 * B * H * H * Cout * k * k * Cin
        out = []
        for bm, bn in [(96, 64), (128, 64), (96, 128), (64, 64), (64, 128), (128, 128)]:
            for on in (0, 1):
                L.lib.call('simq_tune_tail_split', on)
                L.lib.call('simq_tune_force_tile', bm, bn)
                ms = timeit(lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, pad, None, st))
                out.append('%dx%d%s:%.1f' % (bm, bn, '+' if on else '-', flops / ms / 1e9))
        L.lib.call('simq_tune_force_tile', 0, 0)
        print('B=%d %-7s ' % (B, name) + '  '.join(out))
