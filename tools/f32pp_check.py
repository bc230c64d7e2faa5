#!/usr/bin/env python3
"""GPU: time of the batched transform-domain GEMM inside simq_conv2d_fwd_winograd / winograd4 on the fp32 Winograd layer shapes
(HIP events around the GEMM launch only, simq_profile_*).  Run once per kernel choice / ablation:
  SIMQ_F32_PP=0|1|2  SIMQ_F32_PP_DBG=1|2|8|16   usage: f32pp_check.py [B ...]   (default 32)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L

SHAPES = {'l4': (512, 512), 'l4a': (256, 512), 'l4b': (512, 256), 'l3': (256, 256), 'l3a': (128, 256)}
st = L.stream_ptr()


def run(B, name, f4, reps=10):
    H = 24
    Cin, Cout = SHAPES[name]
    g = torch.Generator(device='cpu').manual_seed(7)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda()
    y = torch.empty(B, H, H, Cout, device='cuda')
    T = B * (H // 4) ** 2 if f4 else B * (H // 2) ** 2
    nb = 36 if f4 else 16
    scratch = torch.empty(nb * Cout * Cin + nb * T * (Cin + Cout) + 64, device='cuda')
    fn = 'simq_conv2d_fwd_winograd4' if f4 else 'simq_conv2d_fwd_winograd'
    call = lambda: L.lib.call(fn, L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, None, L.ptr(scratch), st)
    call(); torch.cuda.synchronize()
    L.lib.call('simq_profile_start')
    for _ in range(reps):
        call()
    o = (ctypes.c_double * 12)()
    L.lib.call('simq_profile_stop', o, 3)
    ms = o[1] / max(o[0], 1)
    return ms, 2.0 * T * Cin * Cout * nb


for B in [int(a) for a in sys.argv[1:]] or [32]:
    for f4 in (False, True):
        row = []
        for name in SHAPES:
            ms, fl = run(B, name, f4)
            row.append('%s %.1f us %.1f TF/s' % (name, ms * 1e3, fl / ms / 1e9))
        print('B=%d %s | ' % (B, 'F4' if f4 else 'F2') + ' | '.join(row), flush=True)
