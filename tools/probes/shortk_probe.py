#!/usr/bin/env python3
"""How much of the fp32 implicit-GEMM kernel's rate is lost to short K (prologue / epilogue per block): the same M x N
problem as 1x1 convolutions with K = 128 ... 4096, 64x64 and 96x64 tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L
st = L.stream_ptr()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, H, Cout = 128, 24, 512
for K in (128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(B, H, H, K, device='cuda'); w = torch.randn(Cout, 1, 1, K, device='cuda') * 0.05
    y = torch.empty(B, H, H, Cout, device='cuda')
    out = []
    for tile in ((64, 64), (96, 64), (128, 64), (96, 128)):
        o = L.launch_opts(tile=tile)
        ms = timeit(lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, K, Cout, 1, 1, 1, 0, None, st, opts=o))
        out.append('%dx%d %.1f TF' % (tile[0], tile[1], 2.0 * B * H * H * K * Cout / ms / 1e9))
    print('K=%4d  ' % K + '  '.join(out))
