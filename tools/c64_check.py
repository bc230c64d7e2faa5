#!/usr/bin/env python3
"""GPU: the resident-operand bf16 kernel of the 64-input-channel 3x3 layers (conv_igemm_bf16_c64.hip) against the LDS-DMA 144x64 tile
it replaces: max deviation of the outputs / BN statistics and per-launch times.   usage: c64_check.py [B ...]   (default 128 100)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L

st = L.stream_ptr()


def run(B, Cout, tile, reps=20):
    H, Cin, k = 24, 64, 3
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    y = torch.empty(B, H, H, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    opts = L.launch_opts(tile=tile)
    try:
        L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, 1, L.ptr(scratch), L.ptr(stats), st, opts=opts)
        torch.cuda.synchronize()
        out, s = y.clone(), stats.clone()
        L.lib.call('simq_profile_start')
        for _ in range(reps):
            L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, 1, L.ptr(scratch), None, st, opts=opts)
        o = (ctypes.c_double * 12)()
        L.lib.call('simq_profile_stop', o, 3)
    finally:
        pass
    return out, s, (o[1] + o[9]) / max(o[0] + o[8], 1)


for B in [int(a) for a in sys.argv[1:]] or [128, 100]:
    for Cout in (64, 128):
        new, snew, ms_new = run(B, Cout, (288, 64))
        old, sold, ms_old = run(B, Cout, (144, 64))
        fl = 2.0 * B * 576 * Cout * 576
        err = float((new - old).abs().max() / old.abs().max())
        serr = float((snew - sold).abs().max() / sold.abs().max())
        print('B=%d 64->%d  resident 288x64 %.1f us (%.0f TF/s) | LDS-DMA 144x64 %.1f us (%.0f TF/s)   max dev y %.2e stats %.2e'
              % (B, Cout, ms_new * 1e3, fl / ms_new / 1e9, ms_old * 1e3, fl / ms_old / 1e9, err, serr), flush=True)
        assert err < 1e-4 and serr < 1e-6, (err, serr)
print('c64_check OK')
