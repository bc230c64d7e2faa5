#!/usr/bin/env python3
"""GPU: the bf16 launchers' tiles on the 1x1 convolutions (downsample / head, forward and dgrad geometry) and the narrow 3x3 layers at the
bench's batch (128; 115 = the non-final next states): us per forced tile against the launcher's own choice, with the HBM time of the
launch's operands (bf16 in / out at 5 TB/s) beside it.  Includes the fused BatchNorm statistics (d_stats) as the plan's forwards have them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L

TILES = [(288, 256), (288, 128), (144, 128), (288, 64), (144, 64), (128, 128), (96, 128), (128, 64), (96, 64), (64, 64)]
SHAPES = [('ds4 256->512', 256, 512, 1), ('dg_ds4 512->256', 512, 256, 1), ('h1 512->128', 512, 128, 1), ('dg_h1 128->512', 128, 512, 1),
          ('ds3 128->256', 128, 256, 1), ('dg_ds3 256->128', 256, 128, 1), ('ds2 64->128', 64, 128, 1), ('dg_ds2 128->64', 128, 64, 1),
          ('l2 128->128 3x3', 128, 128, 3), ('l2a 64->128 3x3', 64, 128, 3), ('l1 64->64 3x3', 64, 64, 3)]
st = L.stream_ptr()


def timeit(fn, iters=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


xx = torch.randn(4096, 4096, device='cuda')
for _ in range(60):
    xx @ xx
torch.cuda.synchronize()
H = 24
for B in [int(a) for a in sys.argv[1:]] or [128, 115]:
    for name, Cin, Cout, k in SHAPES:
        x = torch.randn(B, H, H, Cin, device='cuda').relu_(); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
        y = torch.empty(B, H, H, Cout, device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
        # (the entry point splits the fp32 inputs into bf16 planes on every call: time that alone and subtract)
        call = lambda o: L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1, L.ptr(scratch),
                                    L.ptr(stats), st, opts=o)
        L.lib.call('simq_launch_counts_reset')
        auto = timeit(lambda: call(None))
        fam = ','.join(kf for kf in L.launch_counts() if kf.startswith('igemm'))
        res = []
        for t in TILES:
            if Cout % t[1]:
                continue
            try:
                res.append((timeit(lambda: call(L.launch_opts(tile=t))), t))
            except Exception:
                pass
        res.sort()
        hbm = 2.0 * B * H * H * (Cin + Cout) / 5e6
        print('B=%d %-18s auto %6.1f us [%s] (operands at 5 TB/s: %4.1f us) | ' % (B, name, auto, fam, hbm) + '  '.join('%dx%d %.1f' % (t[0], t[1], us) for us, t in res[:5]), flush=True)
print('(every figure includes the fp32 -> bf16 plane split of x and w the entry point performs per call: ~ x bytes * 1.5 / 5 TB/s)')
