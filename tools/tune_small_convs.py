#!/usr/bin/env python3
"""GPU: the fp32 implicit-GEMM tile menu on the SHORT launches of the fp32 step (1x1 convolutions and their dgrads, the 64 -> 128 3x3 layer) at
B = 32 / 29: us per tile, the launcher's own choice marked.  The cost model of launch_conv_igemm was fitted on the long 512-channel 3x3 layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L

TILES = [(128, 128), (96, 128), (64, 128), (128, 64), (96, 64), (64, 64), (128, 32), (96, 32), (64, 32), (32, 64), (32, 32)]
SHAPES = [('h1 512->128', 24, 512, 128, 1, 0), ('dg_h1 128->512', 24, 128, 512, 1, 0), ('h2 128->32', 24, 128, 32, 1, 0), ('dg_h2 32->128', 24, 32, 128, 1, 0),
          ('ds2 64->128', 24, 64, 128, 1, 0), ('dg_ds2 128->64', 24, 128, 64, 1, 0), ('ds3 128->256', 24, 128, 256, 1, 0), ('dg_ds3 256->128', 24, 256, 128, 1, 0),
          ('ds4 256->512', 24, 256, 512, 1, 0), ('dg_ds4 512->256', 24, 512, 256, 1, 0), ('l2a 64->128 3x3', 24, 64, 128, 3, 1), ('dg_l2a 128->64 3x3', 24, 128, 64, 3, 1)]
st = L.stream_ptr()


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# clocks up first
xx = torch.randn(4096, 4096, device='cuda')
for _ in range(50):
    xx @ xx
torch.cuda.synchronize()
for B in [int(a) for a in sys.argv[1:]] or [32, 29]:
    for name, H, Cin, Cout, k, pad in SHAPES:
        x = torch.randn(B, H, H, Cin, device='cuda').relu_(); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
        y = torch.empty(B, H, H, Cout, device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        call = lambda o: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, pad, L.ptr(stats), st, opts=o)
        auto = timeit(lambda: call(None))
        res = []
        for bm, bn in TILES:
            if Cout % bn:
                continue
            o = L.launch_opts(tile=(bm, bn))
            res.append((timeit(lambda: call(o)), bm, bn))
        res.sort()
        fl = 2.0 * B * H * H * Cout * k * k * Cin
        print('B=%d %-20s auto %6.1f us (%5.1f TF) | best ' % (B, name, auto, fl / auto / 1e6) + '  '.join('%dx%d %.1f' % (bm, bn, t) for t, bm, bn in res[:5]), flush=True)
