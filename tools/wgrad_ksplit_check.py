#!/usr/bin/env python3
"""GPU box: launch time of the transform-domain weight gradient (simq_conv2d_wgrad_winograd, F(4x4,3x3)) with the K-split forced to
1 / 2 / 4 and chosen by shape, per layer shape of the network at the batch sizes of the bench workloads.
usage: python tools/wgrad_ksplit_check.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
import torch  # noqa: E402
from simq import _lib as L  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B in (32, 64, 128):
    for Cin, Cout in ((128, 256), (256, 256), (256, 512), (512, 512)):
        x = torch.randn(B, 24, 24, Cin, device='cuda')
        dy = torch.randn(B, 24, 24, Cout, device='cuda')
        dw = torch.empty(Cout, 3, 3, Cin, device='cuda')
        scratch = torch.empty(36 * Cout * Cin + 16 * B * 144 * (Cin + Cout), device='cuda')
        st = L.stream_ptr()
        row = []
        for s in (1, 2, 4, 0):
            o = L.launch_opts(wgrad_ksplit=s)
            row.append(timed(lambda: L.lib.call('simq_conv2d_wgrad_winograd', L.ptr(x), L.ptr(dy), L.ptr(dw), B, 24, 24, Cin, Cout, L.ptr(scratch), st, opts=o)))
        print('B=%3d %3d->%3d  whole launch sequence, us:  S=1 %7.1f   S=2 %7.1f   S=4 %7.1f   auto %7.1f' % ((B, Cin, Cout) + tuple(row)), flush=True)
