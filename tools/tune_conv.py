#!/usr/bin/env python3
"""GPU tuning aid: time every convolution shape of the Q-network (forward and dgrad geometry) at batch B for
every implicit-GEMM tile of the menu, and the wgrad kernel, through the C-ABI.  Prints TFLOP/s per (shape, tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
TILES = [(128, 128), (96, 128), (64, 128), (128, 64), (96, 64), (64, 64), (128, 32), (96, 32), (64, 32), (32, 64), (32, 32)]
# (name, H, Cin, Cout, k, stride, pad)
SHAPES = [('l1', 24, 64, 64, 3, 1, 1), ('l2a', 24, 64, 128, 3, 1, 1), ('l2', 24, 128, 128, 3, 1, 1), ('l3a', 24, 128, 256, 3, 1, 1),
          ('l3', 24, 256, 256, 3, 1, 1), ('l4a', 24, 256, 512, 3, 1, 1), ('l4', 24, 512, 512, 3, 1, 1),
          ('ds4', 24, 256, 512, 1, 1, 0), ('h1', 24, 512, 128, 1, 1, 0), ('h2', 48, 128, 32, 1, 1, 0), ('stem', 96, 4, 64, 7, 2, 3),
          ('dg_l4a', 24, 512, 256, 3, 1, 1), ('dg_l3a', 24, 256, 128, 3, 1, 1), ('dg_h1', 24, 128, 512, 1, 1, 0), ('dg_h2', 48, 32, 128, 1, 1, 0)]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


st = L.stream_ptr()
for name, H, Cin, Cout, k, stride, pad in SHAPES:
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    y = torch.empty(B, Ho, Ho, Cout, device='cuda'); dy = torch.randn(B, Ho, Ho, Cout, device='cuda'); dw = torch.empty_like(w)
    flops = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    res = []
    for bm, bn in TILES:
        if Cout % bn or (Cin % 16 and not (bn == 64 and bm in (128, 64, 32))):
            continue
        o = L.launch_opts(tile=(bm, bn))
        ms = timeit(lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, stride, pad, None, st, opts=o))
        res.append((flops / ms / 1e9, bm, bn, ms))
    ms_auto = timeit(lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, stride, pad, None, st))
    ms_wg = timeit(lambda: L.lib.call('simq_conv2d_wgrad', L.ptr(x), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, k, k, stride, pad, st))
    res.sort(reverse=True)
    print('%-7s M=%6d N=%4d K=%5d  auto %6.1f TF (%.3f ms) | wgrad %6.1f TF (%.3f ms) | ' % (name, B * Ho * Ho, Cout, k * k * Cin, flops / ms_auto / 1e9, ms_auto, flops / ms_wg / 1e9, ms_wg)
          + '  '.join('%dx%d:%.1f' % (bm, bn, tf) for tf, bm, bn, _ in res))

print('--- bf16 matrix-core paths (algorithmic TFLOP/s = 2MNK / time; split-bf16 issues 3 MFMAs per product) ---')
for name, H, Cin, Cout, k, stride, pad in SHAPES:
    if Cin % 32 or stride != 1:
        continue
    x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    y = torch.empty(B, H, H, Cout, device='cuda'); dy = torch.randn(B, H, H, Cout, device='cuda'); dw = torch.empty_like(w)
    scratch = torch.empty(2 * (x.numel() + max(w.numel(), dy.numel())) + 64, dtype=torch.int16, device='cuda')
    flops = 2.0 * B * H * H * Cout * k * k * Cin
    # the entry points re-split the fp32 inputs on every call: time the splitters alone and subtract
    out = []
    for npl in (2, 1):
        ms_f = timeit(lambda: L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, pad, npl, L.ptr(scratch), None, st))
        if Cin % 64 and not (Cout % 32 == 0 and Cin % 128 == 0): continue
        ms_w = timeit(lambda: L.lib.call('simq_conv2d_wgrad_bf16', L.ptr(x), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, k, k, 1, pad, npl, L.ptr(scratch), st))
        out.append('np=%d fwd %.3f ms (%.0f TF incl. split)  wgrad %.3f ms (%.0f TF incl. split)' % (npl, ms_f, flops / ms_f / 1e9, ms_w, flops / ms_w / 1e9))
    print('%-7s M=%6d N=%4d K=%5d  ' % (name, B * H * H, Cout, k * k * Cin) + ' | '.join(out))
