#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace`: the F(4x4,3x3) convolution operator (input transform, 36 batched GEMMs, output transform with
BatchNorm statistics) on the step's own shapes, so that the transform kernels can be read per (kernel, grid) from the trace
(tools/wino4f_trace.py; recipe `w4f`).   usage: wino4f_probe.py [B ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch  # noqa: E402
from simq import _lib as L  # noqa: E402
st = L.stream_ptr()
H = 24
for B in [int(a) for a in sys.argv[1:]] or [32, 29]:
    for cin, cout in [(512, 512), (256, 512), (512, 256), (256, 256), (128, 256), (256, 128), (128, 128)]:
        x = torch.randn(B, H, H, cin, device='cuda')
        w = torch.randn(cout, 3, 3, cin, device='cuda') * (cin * 9) ** -0.5
        b = torch.randn(cout, device='cuda')
        y = torch.empty(B, H, H, cout, device='cuda')
        stats = torch.zeros(2 * cout, dtype=torch.float64, device='cuda')
        T = B * (H // 2) ** 2
        scratch = torch.empty(36 * cout * cin + 16 * T * (cin + cout), device='cuda')
        for _ in range(12):
            L.lib.call('simq_conv2d_fwd_winograd4', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, H, H, cin, cout, L.ptr(stats), L.ptr(scratch), st, None)
        torch.cuda.synchronize()
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
        err = float((y - ref).abs().max() / ref.abs().max())
        s_ref = torch.cat([ref.sum((0, 1, 2)), (ref * ref).sum((0, 1, 2))]) * 12
        s_mag = torch.cat([ref.abs().sum((0, 1, 2)), (ref * ref).sum((0, 1, 2))]) * 12      # a sum's error against the sum of magnitudes
        serr = float(((stats - s_ref).abs() / s_mag).max())
        print('B=%d %d->%d  max err vs fp64 %.2e  statistics %.2e' % (B, cin, cout, err, serr))
        assert err < 2e-5 and serr < 1e-5, (err, serr)
