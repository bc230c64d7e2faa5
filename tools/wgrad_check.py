#!/usr/bin/env python3
"""GPU: the bf16 weight-gradient kernels of the wide 3x3 layers -- time (HIP events through simq_profile_*) and error against the fp64
gradient of the bf16-rounded operands.  Runs on the ablation build: SIMQ_BF16_WGRAD_IMG=0 falls back to the per-tap 256 x 256 ping-pong
tile, SIMQ_BF16_WGRAD_PP=0 further to the register-staged 128 x 128 kernel.   usage: wgrad_check.py [B ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L
st = L.stream_ptr()
tag = 'SLAB=%s IMG=%s PP=%s' % (os.environ.get('SLAB', '1'), os.environ.get('SIMQ_BF16_WGRAD_IMG', '1'), os.environ.get('SIMQ_BF16_WGRAD_PP', '1'))
for B in [int(a) for a in sys.argv[1:]] or [128]:
    for name, (Cin, Cout) in {'l4': (512, 512), 'l4a': (256, 512), 'l3': (256, 256), 'l3a': (128, 256)}.items():
        g = torch.Generator().manual_seed(3 + Cin + Cout + B)
        x = torch.randn(B, 24, 24, Cin, generator=g).cuda(); dy = torch.randn(B, 24, 24, Cout, generator=g).cuda()
        dw = torch.empty(Cout, 3, 3, Cin, device='cuda')
        scratch = torch.empty(2 * (x.numel() + dy.numel()) + 64, dtype=torch.int16, device='cuda')
        slab = torch.empty(L._c.simq_conv2d_wgrad_bf16_slab_bytes() // 4, device='cuda') if os.environ.get('SLAB', '1') != '0' else None
        call = lambda: L.lib.call('simq_conv2d_wgrad_bf16_slab', L.ptr(x), L.ptr(dy), L.ptr(dw), B, 24, 24, Cin, Cout, 3, 3, 1, 1, 1, L.ptr(scratch), L.ptr(slab), st)
        for _ in range(2):
            call()
        err = float('nan')
        if B <= 32 or name == 'l3':
            ref = torch.nn.grad.conv2d_weight(x.bfloat16().double().permute(0, 3, 1, 2), (Cout, Cin, 3, 3), dy.bfloat16().double().permute(0, 3, 1, 2),
                                              padding=1).permute(0, 2, 3, 1)
            err = float((dw.double() - ref).abs().max() / ref.abs().max())
        L.lib.call('simq_profile_start')
        for _ in range(5):
            call()
        o = (ctypes.c_double * 12)()
        L.lib.call('simq_profile_stop', o, 3)
        ms = o[5] / max(o[4], 1)
        fl = 2.0 * B * 576 * Cout * 9 * Cin
        print('%s B=%d %-3s wgrad %.1f us (%.0f TF/s)  max err vs fp64 of the rounded operands %.2g' % (tag, B, name, ms * 1e3, fl / ms / 1e9, err), flush=True)
