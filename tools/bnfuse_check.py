#!/usr/bin/env python
"""What the BatchNorm-on-load costs its consumers (fp32 plans, simq_plan_options.fuse_bn1_apply): per shape the time of the fused call
conv(relu(bn(y))) / its weight gradient against the plain call on a materialised activation, alone on the device (GPU box), and the
time a separate elementwise pass over the same map costs at 5.5 TB/s (what the fusion removes: one read + one write of the map).
Both sides include the per-call weight transform of the C-ABI test entry points (a constant)."""
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


def one(B, Cin, Cout, k, form, H=24):
    pad = k // 2
    y = torch.randn(B, H, H, Cin, device='cuda')
    sc = torch.rand(Cin, device='cuda') + 0.5
    sh = torch.randn(Cin, device='cuda') * 0.5
    a = torch.relu(y * sc + sh)
    w = torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5
    out = torch.empty(B, H, H, Cout, device='cuda')
    dy = torch.randn(B, H, H, Cout, device='cuda')
    dw = torch.empty(Cout, k, k, Cin, device='cuda')
    T = B * (H // 2) ** 2
    scratch = torch.empty(36 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda')
    st = L.stream_ptr()
    fused = lambda: L.lib.call('simq_conv2d_fwd_bnrelu_in', L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(w), None, L.ptr(out), B, H, H, Cin, Cout, k, k, 1, pad,
                               form, L.ptr(scratch), st)
    if form == 0:
        plain = lambda: L.lib.call('simq_conv2d_fwd', L.ptr(a), L.ptr(w), None, L.ptr(out), B, H, H, Cin, Cout, k, k, 1, pad, None, st)
    elif form == 1:
        plain = lambda: L.lib.call('simq_conv2d_fwd_winograd', L.ptr(a), L.ptr(w), None, L.ptr(out), B, H, H, Cin, Cout, None, L.ptr(scratch), st)
    else:
        plain = lambda: L.lib.call('simq_conv2d_fwd_winograd4', L.ptr(a), L.ptr(w), None, L.ptr(out), B, H, H, Cin, Cout, None, L.ptr(scratch), st)
    tf, tp = timeit(fused), timeit(plain)
    wform = 1 if (form and Cin % 128 == 0 and Cout % 128 == 0 and Cin * Cout >= 128 * 256) else 0
    wf = lambda: L.lib.call('simq_conv2d_wgrad_bnrelu_in', L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, k, k, 1, pad, wform,
                            L.ptr(scratch), st)
    if wform:
        wp = lambda: L.lib.call('simq_conv2d_wgrad_winograd', L.ptr(a), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, L.ptr(scratch), st)
    else:
        wp = lambda: L.lib.call('simq_conv2d_wgrad', L.ptr(a), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, k, k, 1, pad, st)
    twf, twp = timeit(wf), timeit(wp)
    pass_us = 2.0 * B * H * H * Cin * 4 / 5.5e12 * 1e6
    print('B=%3d %3d->%3d k%d form %d: forward fused %7.1f us | plain %7.1f us (+%5.1f) ; wgrad(form %d) fused %7.1f | plain %7.1f (+%5.1f) ; a separate pass over the map: %5.1f us'
          % (B, Cin, Cout, k, form, tf, tp, tf - tp, wform, twf, twp, twf - twp, pass_us))


for B in (32, 29):
    one(B, 64, 64, 3, 0)
    one(B, 128, 128, 3, 1)
    one(B, 128, 128, 3, 2)
    one(B, 256, 256, 3, 1)
    one(B, 256, 256, 3, 2)
    one(B, 512, 512, 3, 1)
    one(B, 512, 512, 3, 2)
    one(B, 128, 32, 1, 0)
