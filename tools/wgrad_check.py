#!/usr/bin/env python3
"""GPU: time the bf16 weight-gradient kernels on the wide layers (HIP events through simq_profile_*): SIMQ_BF16_WGRAD_PP=0 keeps the
register-staged 128x128 kernel.  usage: wgrad_check.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
st = L.stream_ptr()
for name, (Cin, Cout) in {'l4': (512, 512), 'l4a': (256, 512), 'l3': (256, 256)}.items():
    x = torch.randn(B, 24, 24, Cin, device='cuda'); dy = torch.randn(B, 24, 24, Cout, device='cuda')
    dw = torch.empty(Cout, 3, 3, Cin, device='cuda')
    scratch = torch.empty(2 * (x.numel() + dy.numel()) + 64, dtype=torch.int16, device='cuda')
    for _ in range(2):
        L.lib.call('simq_conv2d_wgrad_bf16', L.ptr(x), L.ptr(dy), L.ptr(dw), B, 24, 24, Cin, Cout, 3, 3, 1, 1, 1, L.ptr(scratch), st)
    L.lib.call('simq_profile_start')
    for _ in range(5):
        L.lib.call('simq_conv2d_wgrad_bf16', L.ptr(x), L.ptr(dy), L.ptr(dw), B, 24, 24, Cin, Cout, 3, 3, 1, 1, 1, L.ptr(scratch), st)
    o = (ctypes.c_double * 12)()
    L.lib.call('simq_profile_stop', o, 3)
    ms = o[5] / max(o[4], 1)
    fl = 2.0 * B * 576 * Cout * 9 * Cin
    print('PP=%s B=%d %-3s wgrad %.1f us (%.0f TF/s)' % (os.environ.get('SIMQ_BF16_WGRAD_PP', '1'), B, name, ms * 1e3, fl / ms / 1e9), flush=True)
