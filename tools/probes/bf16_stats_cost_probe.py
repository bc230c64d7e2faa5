import os, sys
ROOT = '/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L
st = L.stream_ptr()
def timeit(fn, iters=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
xx = torch.randn(4096, 4096, device='cuda')
for _ in range(60): xx @ xx
torch.cuda.synchronize()
H, B = 24, 128
for name, Cin, Cout, k in [('ds4 256->512', 256, 512, 1), ('h1 512->128', 512, 128, 1), ('ds3 128->256', 128, 256, 1), ('ds2 64->128', 64, 128, 1), ('l1 64->64 3x3', 64, 64, 3), ('l2 128->128 3x3', 128, 128, 3), ('l4 512->512 3x3', 512, 512, 3)]:
    x = torch.randn(B, H, H, Cin, device='cuda').relu_(); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    y = torch.empty(B, H, H, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    f = lambda s: L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1, L.ptr(scratch), s, st)
    hi, lo = torch.empty(x.numel(), dtype=torch.int16, device='cuda'), None
    split = timeit(lambda: (x.to(torch.bfloat16), w.to(torch.bfloat16)))
    print('%-18s with stats %6.1f us   without %6.1f us   (torch fp32->bf16 casts of x, w alone: %.1f us)' % (name, timeit(lambda: f(L.ptr(stats))), timeit(lambda: f(None)), split), flush=True)
