#!/usr/bin/env python3
"""GPU: what the BatchNorm-statistics epilogue (fp64 atomics, one per channel and block) costs the small fp32 convolutions:
simq_conv2d_fwd with and without d_stats on the layer1 / layer2 shapes at B = 32 (HIP events around 20 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L
st = L.stream_ptr()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for name, (Cin, Cout, k) in {'l1 64->64': (64, 64, 3), 'l2a 64->128': (64, 128, 3), 'l2 128->128': (128, 128, 3), 'ds 64->128 1x1': (64, 128, 1), 'l3a 128->256': (128, 256, 3)}.items():
    x = torch.randn(B, 24, 24, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    y = torch.empty(B, 24, 24, Cout, device='cuda'); stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    res = []
    for s in (None, stats):
        for _ in range(3):
            L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, 24, 24, Cin, Cout, k, k, 1, k // 2, L.ptr(s) if s is not None else None, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, 24, 24, Cin, Cout, k, k, 1, k // 2, L.ptr(s) if s is not None else None, st)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * B * 576 * Cout * k * k * Cin
    print('%-16s B=%d  plain %.1f us (%.0f TF/s)   with statistics %.1f us (%.0f TF/s)' % (name, B, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6), flush=True)
