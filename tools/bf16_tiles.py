#!/usr/bin/env python3
"""GPU tuning aid: run the bf16 implicit-GEMM forward of one layer shape under every tile of the menus (kernel names
carry the tile, so `rocprofv3 --kernel-trace` + tools/rocprof_summary.py gives the per-tile kernel time; tools/bt.sh wraps
that).  BF16_DBGS="1 2 8 16 23 32 100" additionally runs the timing ablations of the LDS-DMA kernel (SIMQ_BF16_DBG bit mask:
1 no DMA, 2 no barrier, 8 no fragment reads, 16 no MFMA, 23 fragment reads only, 32 epilogue only; 100 = four-wave variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
# SIMQ_* kernel-selection / ablation switches exist in the ablation build only (make -C spatial-intention-maps_amd/csrc ablate)
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch
from simq import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = {'l4': (24, 512, 512, 3), 'l3': (24, 256, 256, 3), 'l2': (24, 128, 128, 3), 'l1': (24, 64, 64, 3)}
TILES = [(288, 128), (144, 128), (288, 64), (144, 64), (128, 128), (96, 128), (96, 64)]
st = L.stream_ptr()
for name in sys.argv[2:] or ['l4']:
    H, Cin, Cout, k = SHAPES[name]
    x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    y = torch.empty(B, H, H, Cout, device='cuda')
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    flops = 2.0 * B * H * H * Cout * k * k * Cin
    for bm, bn in TILES:
        if Cout % bn:
            continue
        for _ in range(4):
            L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1, L.ptr(scratch), None, st,
                       opts=L.launch_opts(tile=(bm, bn)))
        torch.cuda.synchronize()
    for dbg in os.environ.get('BF16_DBGS', '').split():
        os.environ['SIMQ_BF16_DBG'] = dbg
        for _ in range(4):
            L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1, L.ptr(scratch), None, st,
                       opts=L.launch_opts(tile=(288, 128)))
        torch.cuda.synchronize()
    os.environ.pop('SIMQ_BF16_DBG', None)
    print(name, 'GFLOP', flops / 1e9)
