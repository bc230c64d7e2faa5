#!/usr/bin/env python3
"""GPU tuning aid: Winograd F(2x2,3x3) path vs the implicit-GEMM kernel on the 3x3 layer shapes (time and difference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L
st = L.stream_ptr()


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B in [int(a) for a in sys.argv[1:]] or [32, 29]:
    for name, Cin, Cout in [('l4', 512, 512), ('l4a', 256, 512), ('dg_l4a', 512, 256), ('l3', 256, 256), ('l3a', 128, 256), ('dg_l3a', 256, 128), ('l2', 128, 128), ('l2a', 64, 128), ('l1', 64, 64)]:
        H = 24
        x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, 3, 3, Cin, device='cuda') * (Cin * 9) ** -0.5
        b = torch.randn(Cout, device='cuda')
        T = B * (H // 2) ** 2
        scratch = torch.empty(16 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda')
        y0, y1 = torch.empty(B, H, H, Cout, device='cuda'), torch.empty(B, H, H, Cout, device='cuda')
        s0, s1 = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda'), torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y0), B, H, H, Cin, Cout, 3, 3, 1, 1, L.ptr(s0), st)
        L.lib.call('simq_conv2d_fwd_winograd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y1), B, H, H, Cin, Cout, L.ptr(s1), L.ptr(scratch), st)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
        sc = float(ref.abs().max())
        e0, e1 = float((y0 - ref).abs().max()) / sc, float((y1 - ref).abs().max()) / sc
        es = float((s1 - s0).abs().max() / s0.abs().max())
        md = timeit(lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y0), B, H, H, Cin, Cout, 3, 3, 1, 1, None, st))
        mw = timeit(lambda: L.lib.call('simq_conv2d_fwd_winograd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y1), B, H, H, Cin, Cout, None, L.ptr(scratch), st))
        flops = 2.0 * B * H * H * Cout * 9 * Cin
        print('B=%3d %-7s direct %.3f ms (%.1f TF)  winograd %.3f ms (%.1f TF effective, incl. weight transform)  x%.2f | err vs fp64: direct %.1e winograd %.1e  stats diff %.1e'
              % (B, name, md, flops / md / 1e9, mw, flops / mw / 1e9, md / mw, e0, e1, es))

print('--- weight gradient: direct kernel vs transform domain ---')
for B in (32,):
    for name, Cin, Cout in [('l4', 512, 512), ('l4a', 256, 512), ('l3', 256, 256), ('l3a', 128, 256), ('l2', 128, 128)]:
        H = 24
        x = torch.randn(B, H, H, Cin, device='cuda'); dy = torch.randn(B, H, H, Cout, device='cuda')
        T = B * (H // 2) ** 2
        scratch = torch.empty(36 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda')
        d0, d1 = torch.empty(Cout, 3, 3, Cin, device='cuda'), torch.empty(Cout, 3, 3, Cin, device='cuda')
        f0 = lambda: L.lib.call('simq_conv2d_wgrad', L.ptr(x), L.ptr(dy), L.ptr(d0), B, H, H, Cin, Cout, 3, 3, 1, 1, st)
        f1 = lambda: L.lib.call('simq_conv2d_wgrad_winograd', L.ptr(x), L.ptr(dy), L.ptr(d1), B, H, H, Cin, Cout, L.ptr(scratch), st)
        m0, m1 = timeit(f0), timeit(f1)
        xd = x.permute(0, 3, 1, 2).double().requires_grad_(False)
        wref = torch.nn.grad.conv2d_weight(xd, (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
        sc = float(wref.abs().max())
        flops = 2.0 * B * H * H * Cout * 9 * Cin
        print('B=%3d %-5s direct %.3f ms (%.1f TF)  winograd %.3f ms (%.1f TF effective)  x%.2f | err vs fp64: direct %.1e winograd %.1e  [batched splits %s]'
              % (B, name, m0, flops / m0 / 1e9, m1, flops / m1 / 1e9, m0 / m1, float((d0 - wref).abs().max()) / sc, float((d1 - wref).abs().max()) / sc,
                 os.environ.get('SIMQ_WGRAD_BATCHED_SPLITS', 'auto')))
