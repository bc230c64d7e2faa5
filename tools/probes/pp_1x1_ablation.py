#!/usr/bin/env python3
"""GPU, ablation build (one process per SIMQ_BF16_PP_DBG value): the 288x256 ping-pong kernel on the 1x1 downsample convolution 256 -> 512 at
B = 128 (M = 73 728, K = 256: eight K-tiles per block, 512 blocks = two rounds) -- where do its 60 us go?  usage: pp_1x1_ablation.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
    import torch
    from simq import _lib as L
    st = L.stream_ptr()
    xx = torch.randn(4096, 4096, device='cuda')
    for _ in range(40):
        xx @ xx
    H, B = 24, 128
    for name, Cin, Cout in (('ds4 256->512', 256, 512), ('dg_h1 128->512', 128, 512), ('ds3 128->256', 128, 256)):
        x = torch.randn(B, H, H, Cin, device='cuda').relu_(); w = torch.randn(Cout, 1, 1, Cin, device='cuda') * 0.05
        y = torch.empty(B, H, H, Cout, device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
        f = lambda: L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, 1, 1, 1, 0, 1, L.ptr(scratch), L.ptr(stats), st,
                               opts=L.launch_opts(tile=(288, 256)))
        for _ in range(4):
            f()
        import ctypes
        L.lib.call('simq_profile_start')
        for _ in range(20):
            f()
        o = (ctypes.c_double * 12)()
        L.lib.call('simq_profile_stop', o, 3)
        print('DBG=%-4s %-16s kernel alone %6.1f us' % (os.environ.get('SIMQ_BF16_PP_DBG', '0'), name, (o[1] + o[9]) / max(o[0] + o[8], 1) * 1e3), flush=True)
    sys.exit(0)
for dbg, what in ((0, 'everything'), (32, 'epilogue only'), (16, 'no MFMAs'), (1, 'no DMA'), (64, 'DMA issued, all lanes out of range'), (8, 'no fragment reads')):
    env = dict(os.environ, SIMQ_LIBRARY=os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'), SIMQ_BF16_PP_DBG=str(dbg))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    print('--', what); print(out.strip())
