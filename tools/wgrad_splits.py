#!/usr/bin/env python3
"""GPU tuning aid: fp32 wgrad time of one layer shape as a function of the pixel-reduction split count (SIMQ_WGRAD_SPLITS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = L.stream_ptr()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, H, Cin, Cout, k in [('l4', 24, 512, 512, 3), ('l3', 24, 256, 256, 3), ('l2', 24, 128, 128, 3)]:
    x = torch.randn(B, H, H, Cin, device='cuda'); dy = torch.randn(B, H, H, Cout, device='cuda'); dw = torch.zeros(Cout, k, k, Cin, device='cuda')
    fl = 2.0 * B * H * H * Cout * k * k * Cin
    out = []
    for s in [0] + [int(v) for v in os.environ.get('SPLITS', '3 5 7 10 14 21 28 56').split()]:
        if s: os.environ['SIMQ_WGRAD_SPLITS'] = str(s)
        else: os.environ.pop('SIMQ_WGRAD_SPLITS', None)
        ms = timeit(lambda: L.lib.call('simq_conv2d_wgrad', L.ptr(x), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, k, k, 1, 1, st))
        out.append('%s:%.0fus/%.0fTF' % ('auto' if not s else 's%d' % s, ms * 1e3, fl / ms / 1e9))
    print(name, ' '.join(out))
