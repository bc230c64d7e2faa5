#!/usr/bin/env python3
"""GPU: the 288x256 ping-pong bf16 implicit-GEMM kernel (conv_igemm_bf16_pp.hip) against the register-staged 128x128 tile and the
288x128 LDS-DMA tile on the layer shapes it serves: max deviation of the outputs / BN statistics and HIP-event times.
usage: pp_check.py [B ...]   (default 128 115)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L

SHAPES = {'l2': (24, 128, 128, 3), 'l4': (24, 512, 512, 3), 'l3': (24, 256, 256, 3), 'l3a': (24, 128, 256, 3), 'l4a': (24, 256, 512, 3), 'l4b': (24, 512, 256, 3)}
st = L.stream_ptr()


def run(B, name, tile, reps=5):
    H, Cin, Cout, k = SHAPES[name]
    g = torch.Generator(device='cpu').manual_seed(7)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    y = torch.empty(B, H, H, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    opts = L.launch_opts(tile=tile)
    L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1, L.ptr(scratch), L.ptr(stats), st, opts=opts)
    torch.cuda.synchronize()
    out, s = y.clone(), stats.clone()
    L.lib.call('simq_profile_start')
    for _ in range(reps):
        L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1, L.ptr(scratch), None, st, opts=opts)
    import ctypes
    o = (ctypes.c_double * 12)()
    L.lib.call('simq_profile_stop', o, 3)
    ms = (o[1] + o[9]) / max(o[0] + o[8], 1)
    return out, s, ms, 2.0 * B * H * H * Cout * k * k * Cin


if os.environ.get('SIMQ_BF16_PP_DBG') or os.environ.get('SIMQ_BF16_IMG_DBG'):   # timing ablations (results are wrong by construction)
    for name in ('l4', 'l3'):
        tile = (288, 256) if os.environ.get('SIMQ_BF16_PP_DBG') else ((1288, 128) if os.environ.get('SIMQ_PP_CHECK_HALF') else (576, 128))
        _, _, ms, fl = run(128, name, tile)
        print('DBG=%s B=128 %s %dx%d %.1f us' % (os.environ.get('SIMQ_BF16_PP_DBG') or os.environ.get('SIMQ_BF16_IMG_DBG'), name, tile[0], tile[1], ms * 1e3), flush=True)
    sys.exit(0)
for B in [int(a) for a in sys.argv[1:]] or [128, 115]:
    for name in SHAPES:
        ref, sref, ms_ref, fl = run(B, name, (128, 128))
        dma, sdma, ms_dma, _ = run(B, name, (288, 128))
        pp, spp, ms_pp, _ = run(B, name, (288, 256))
        im, sim, ms_im, _ = run(B, name, (576, 128))
        err = max(float((pp - ref).abs().max() / ref.abs().max()), float((im - ref).abs().max() / ref.abs().max()))
        serr = max(float((spp - sref).abs().max() / sref.abs().max()), float((sim - sref).abs().max() / sref.abs().max()))
        print('B=%d %-3s  128x128 %.1f us (%.0f TF/s) | 288x128 dma %.1f us (%.0f) | 288x256 pp %.1f us (%.0f) | image-tile 576x128 %.1f us (%.0f TF/s)   max dev y %.2e stats %.2e'
              % (B, name, ms_ref * 1e3, fl / ms_ref / 1e9, ms_dma * 1e3, fl / ms_dma / 1e9, ms_pp * 1e3, fl / ms_pp / 1e9, ms_im * 1e3, fl / ms_im / 1e9, err, serr), flush=True)
        assert err < 1e-4 and serr < 1e-6, (err, serr)
print('pp_check OK')
