#!/usr/bin/env python3
"""GPU: round-off of the two forms of the batched transform-domain GEMM against fp64 -- simq_gemm_f32_batched with gemm_split = 0
(v_mfma_f32_16x16x4_f32 on the fp32 operands) and 1 (bf16 matrix cores, exact three-way operand split, six partial products) -- on
operands shaped like the Winograd planes (random normal; and a wide-dynamic-range case).  Prints max / rms error relative to the rms of
the exact result, and the ratio split / fp32-MFMA.   usage: tools/gemm_split_accuracy.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
from simq import _lib as L
st = L.stream_ptr()
torch.manual_seed(0)
for name, M, N, K, P, kind in (('normal', 1152, 512, 512, 36, 0), ('normal', 1044, 256, 128, 36, 0), ('normal', 512, 512, 1152, 36, 0),
                               ('normal', 4176, 128, 128, 16, 0), ('wide range', 1152, 256, 256, 8, 1), ('one-signed', 1152, 256, 256, 8, 2)):
    x = torch.randn(P, M, K, device='cuda'); w = torch.randn(P, N, K, device='cuda')
    if kind == 1:
        x = x * torch.exp2(torch.randint(-20, 20, x.shape, device='cuda').float()); w = w * torch.exp2(torch.randint(-20, 20, w.shape, device='cuda').float())
    if kind == 2:
        x, w = x.abs(), w.abs()
    ref = torch.bmm(x.double(), w.double().transpose(1, 2))
    scale = float(ref.pow(2).mean().sqrt())
    out = {}
    for split in (0, 1):
        y = torch.full((P, M, N), float('nan'), device='cuda')
        L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y), M, N, K, P, st, opts=L.launch_opts(gemm_split=split))
        torch.cuda.synchronize()
        d = (y.double() - ref)
        out[split] = (float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale)
    tm = torch.bmm(x, w.transpose(1, 2)).double() - ref
    print('%-10s M=%5d N=%4d K=%5d x%2d   fp32 MFMA: max %.3e rms %.3e   split3: max %.3e rms %.3e   (split / fp32: max %.2f rms %.2f)   torch.bmm fp32 rms %.3e' % (
        name, M, N, K, P, out[0][0], out[0][1], out[1][0], out[1][1], out[1][0] / out[0][0], out[1][1] / out[0][1], float(tm.pow(2).mean().sqrt()) / scale), flush=True)
