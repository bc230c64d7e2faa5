import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/spatial-intention-maps_amd']
import torch
from simq import _lib as L
st = L.stream_ptr()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for B, Cin, Cout in [(128, 512, 512), (128, 256, 512), (128, 256, 256), (128, 512, 256), (116, 512, 512)]:
    H = 24
    x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, 1, 1, Cin, device='cuda') * 0.05
    y = torch.empty(B, H, H, Cout, device='cuda')
    res = []
    for tile in [(0, 0), (96, 64), (128, 64), (96, 128), (64, 64), (64, 128)]:
        o = L.launch_opts(tile=tile)
        ms = timeit(lambda: L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, H, H, Cin, Cout, 1, 1, 1, 0, None, st, opts=o))
        res.append('%dx%d: %.1f TF (%.3f ms)' % (tile[0], tile[1], 2.0 * B * H * H * Cin * Cout / ms / 1e9, ms))
    print(B, Cin, Cout, ' | '.join(res))
