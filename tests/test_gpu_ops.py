"""GPU: every HIP kernel of libsimq against the same op in plain PyTorch fp32 on the CPU
(conv / dgrad / wgrad / bilinear / argmax / Huber / clip+SGD), through the C-ABI.
Tolerance: 1e-4 relative (max-abs error over max-abs value) -- BASELINE.json's fp32 bar;
observed errors are ~1e-6.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope='module')
def L():
    from simq import _lib
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return _lib


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def dev(t):
    return t.contiguous().cuda()


def nhwc(t):     # NCHW cpu -> NHWC cuda
    return dev(t.permute(0, 2, 3, 1))


def ohwi(w):     # OIHW cpu -> OHWI cuda
    return dev(w.permute(0, 2, 3, 1))


CONV_CASES = [
    # B, H, Cin, Cout, k, stride, pad, bias
    (2, 24, 64, 64, 3, 1, 1, False),
    (3, 24, 128, 256, 3, 1, 1, False),      # M = 1728: ragged last 128-row tile
    (1, 24, 512, 512, 3, 1, 1, False),      # B=1 inference shape (config 0)
    (2, 24, 256, 512, 1, 1, 0, False),      # downsample 1x1
    (2, 24, 512, 128, 1, 1, 0, True),       # head conv1 (+bias)
    (2, 48, 128, 32, 1, 1, 0, True),        # head conv2 (+bias), N = 32 tile
    (2, 96, 4, 64, 7, 2, 3, False),         # stem, generic gather
    (3, 96, 5, 64, 7, 2, 3, False),         # stem, Cin = 5 (K = 245, unaligned rows)
    # the reference's other input-channel counts (tools_generate_experiments.py:200-204): generic-gather forward (what Cin = 10 plans run:
    # the dedicated stem kernel needs 7 * Cin <= 64) and the fp32 stem weight gradient (wgrad_kernel on K = 49 * Cin, unaligned rows)
    (2, 96, 3, 64, 7, 2, 3, False),
    (2, 96, 6, 64, 7, 2, 3, False),
    (3, 96, 7, 64, 7, 2, 3, False),
    (2, 96, 10, 64, 7, 2, 3, False),
    (16, 24, 64, 128, 3, 1, 1, False),      # enough rows for the 128x128 tile path
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'B%d_H%d_%dto%d_k%d' % (c[0], c[1], c[2], c[3], c[4]))
def test_conv_fwd_dgrad_wgrad(L, case):
    B, H, Cin, Cout, k, stride, pad, bias = case
    g = torch.Generator().manual_seed(1234 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    Ho = y_ref.shape[2]
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    st = L.stream_ptr()
    xd, wd = nhwc(x.detach()), ohwi(w.detach())
    bd = dev(b) if bias else None
    # forward (+ per-channel statistics epilogue)
    yd = torch.empty(B, Ho, Ho, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    L.lib.call('simq_conv2d_fwd', L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(yd), B, H, H, Cin, Cout, k, k, stride, pad,
               L.ptr(stats), st)
    y_nhwc = y_ref.detach().permute(0, 2, 3, 1)
    assert rel(yd, y_nhwc) < TOL
    s_ref = y_nhwc.double().reshape(-1, Cout)
    assert rel(stats[:Cout], s_ref.sum(0)) < 1e-5
    assert rel(stats[Cout:], (s_ref * s_ref).sum(0)) < 1e-5
    # weight gradient
    dyd = nhwc(dy)
    dwd = torch.full((Cout, k, k, Cin), 7.0, device='cuda')       # callee zero-fills
    L.lib.call('simq_conv2d_wgrad', L.ptr(xd), L.ptr(dyd), L.ptr(dwd), B, H, H, Cin, Cout, k, k, stride, pad, st)
    assert rel(dwd, w.grad.permute(0, 2, 3, 1)) < TOL
    # data gradient (stride-1 convolutions only; the stem never needs one)
    if stride == 1:
        dxd = torch.empty(B, H, H, Cin, device='cuda')
        wt = torch.empty(Cout * k * k * Cin, device='cuda')
        L.lib.call('simq_conv2d_dgrad', L.ptr(dyd), L.ptr(wd), L.ptr(wt), L.ptr(dxd), B, H, H, Cin, Cout, k, k, pad, st)
        assert rel(dxd, x.grad.permute(0, 2, 3, 1)) < TOL


def test_conv_transpose_detect(L):
    """A = I with an ASYMMETRIC B: catches a swapped row/col in the MFMA C/D mapping."""
    C = 64
    x = torch.zeros(1, C, 24, 24)
    for c in range(C):
        x[0, c, c % 24, (c * 7) % 24] = 1.0
    w = torch.arange(C * C, dtype=torch.float32).reshape(C, C, 1, 1) / 100.0
    y_ref = F.conv2d(x, w)
    yd = torch.empty(1, 24, 24, C, device='cuda')
    xd, wd = nhwc(x), ohwi(w)
    L.lib.call('simq_conv2d_fwd', L.ptr(xd), L.ptr(wd), None, L.ptr(yd), 1, 24, 24, C, C, 1, 1, 1, 0, None,
               L.stream_ptr())
    assert torch.equal(yd.cpu(), y_ref.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize('shape', [(2, 24, 128), (3, 48, 32), (1, 5, 4)])
def test_upsample2x(L, shape):
    B, H, C = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, C, H, H, generator=g, requires_grad=True)
    y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    st = L.stream_ptr()
    yd = torch.empty(B, 2 * H, 2 * H, C, device='cuda')
    xd, dyd = nhwc(x.detach()), nhwc(dy)
    L.lib.call('simq_upsample2x_fwd', L.ptr(xd), L.ptr(yd), B, H, H, C, st)
    assert rel(yd, y.detach().permute(0, 2, 3, 1)) < 1e-6
    dxd = torch.empty(B, H, H, C, device='cuda')
    L.lib.call('simq_upsample2x_bwd', L.ptr(dyd), L.ptr(dxd), B, H, H, C, st)
    assert rel(dxd, x.grad.permute(0, 2, 3, 1)) < 1e-5


def test_layout_helpers(L):
    x = torch.randn(3, 5, 96, 96)
    st = L.stream_ptr()
    out = torch.empty(3, 96, 96, 5, device='cuda')
    xd = dev(x)
    L.lib.call('simq_nchw_to_nhwc', L.ptr(xd), L.ptr(out), 3, 5, 96 * 96, st)
    assert torch.equal(out.cpu(), x.permute(0, 2, 3, 1).contiguous())
    back = torch.empty(3, 5, 96, 96, device='cuda')
    L.lib.call('simq_nhwc_to_nchw', L.ptr(out), L.ptr(back), 3, 5, 96 * 96, st)
    assert torch.equal(back.cpu(), x)


def test_argmax_gather_first_index(L):
    n = 2 * 96 * 96
    q = torch.randn(5, n)
    q[0, 17] = q[0, 9000] = 50.0         # tie -> first index (train.py:121 / policies.py:64 semantics)
    q[1, n - 1] = 60.0
    q[2] = 0.0                           # all equal -> index 0
    st = L.stream_ptr()
    qd = dev(q)
    idx = torch.empty(5, dtype=torch.int64, device='cuda')
    mx = torch.empty(5, device='cuda')
    L.lib.call('simq_q_argmax', L.ptr(qd), 5, n, L.ptr(idx), L.ptr(mx), st)
    ref_max, ref_idx = q.max(1)
    assert idx.cpu().tolist() == ref_idx.tolist()
    assert idx.cpu().tolist()[:3] == [17, n - 1, 0]
    assert torch.equal(mx.cpu(), ref_max)
    out = torch.empty(5, device='cuda')
    L.lib.call('simq_q_gather', L.ptr(qd), 5, n, L.ptr(idx), L.ptr(out), st)
    assert torch.equal(out.cpu(), ref_max)


def test_td_huber_and_scatter(L):
    B, n = 9, 2 * 96 * 96
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, n, generator=g, requires_grad=True)
    a = torch.randint(0, n, (B,), generator=g)
    r = torch.randn(B, generator=g) * 2
    mask = torch.tensor([1, 0, 1, 1, 0, 1, 1, 1, 0], dtype=torch.bool)
    vals = torch.randn(int(mask.sum()), generator=g)
    nsv_ref = torch.zeros(B)
    nsv_ref[mask] = vals
    gamma = 0.85
    q_sa = q.gather(1, a.unsqueeze(1)).squeeze(1)
    y = r + gamma * nsv_ref
    loss = F.smooth_l1_loss(q_sa, y)
    loss.backward()
    st = L.stream_ptr()
    pos = dev(torch.nonzero(mask).squeeze(1).to(torch.int32))
    nsv = torch.full((B,), 9.0, device='cuda')
    vals_d, q_d, a_d, r_d = dev(vals), dev(q.detach()), dev(a), dev(r)   # keep alive: raw pointers cross the ABI
    L.lib.call('simq_scatter_next_values', L.ptr(vals_d), L.ptr(pos), int(mask.sum()), L.ptr(nsv), B, st)
    assert torch.equal(nsv.cpu(), nsv_ref)
    outs = [torch.empty(B, device='cuda') for _ in range(3)]
    out4 = torch.empty(4, device='cuda')
    dq = torch.empty(B, n, device='cuda')
    L.lib.call('simq_td_huber', L.ptr(q_d), B, n, L.ptr(a_d), L.ptr(r_d), L.ptr(nsv), gamma, 1.0 / B,
               L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(out4), L.ptr(dq), st)
    assert rel(outs[0], q_sa) < 1e-6 and rel(outs[1], y) < 1e-6
    assert rel(outs[2], (q_sa - y).abs()) < 1e-6
    assert abs(out4[0].item() / B - loss.item()) < 1e-6 * max(1, abs(loss.item()))
    assert rel(dq, q.grad) < 1e-6
    assert int((dq != 0).sum()) <= B


@pytest.mark.parametrize('B,nf', [(1, 0), (1, 1), (1024, 1024), (1025, 700), (4096, 3700)], ids=['one_terminal', 'one', 'b1024_all', 'b1025', 'b4096_max'])
def test_scatter_next_values_edge_sizes(L, B, nf):
    """next_state_values[non_final_mask] = ... (train.py:116,122) at the edges: no non-final row, every row, and the library's largest minibatch
    (4096 rows: the limit of the forward / backward entry points; the kernel is one block walking the rows) -- and one row more is refused."""
    g = torch.Generator().manual_seed(B + nf)
    pos = torch.randperm(B, generator=g)[:nf].sort().values.to(torch.int32)
    vals = torch.randn(max(nf, 1), generator=g)
    ref = torch.zeros(B)
    ref[pos.long()] = vals[:nf]
    pos_d, vals_d = dev(pos) if nf else torch.zeros(1, dtype=torch.int32, device='cuda'), dev(vals)
    nsv = torch.full((B + 3,), 9.0, device='cuda')
    L.lib.call('simq_scatter_next_values', L.ptr(vals_d), L.ptr(pos_d), nf, L.ptr(nsv), B, L.stream_ptr())
    assert torch.equal(nsv[:B].cpu(), ref) and bool((nsv[B:] == 9.0).all())
    if B == 4096:
        with pytest.raises(Exception, match='unsupported'):
            L.lib.call('simq_scatter_next_values', L.ptr(vals_d), L.ptr(pos_d), nf, L.ptr(nsv), B + 1, L.stream_ptr())


@pytest.mark.parametrize('count,max_norm', [(1003, 100.0), (4096, 0.5), (11249826, 100.0)])
def test_clip_sgd(L, count, max_norm):
    """clip_grad_norm_ + SGD(momentum 0.9, wd 1e-4) (train.py:133-135,186) restated in fp64.
    (torch's own fp32 CPU norm of an 11M-element vector is only ~5e-4 accurate, so the yardstick
    is the same formula in double precision; the small cases are also checked against torch.optim.)"""
    g = torch.Generator().manual_seed(count)
    p = torch.randn(count, generator=g)
    gr = torch.randn(count, generator=g) * 0.05
    p64, m64 = p.double(), None
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pt], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, md = dev(p), torch.zeros(count, device='cuda')
    scratch = torch.zeros(4, dtype=torch.float64, device='cuda')
    tn = torch.zeros(1, device='cuda')
    st = L.stream_ptr()
    for step in range(3):
        grad = gr * (step + 1)
        g64 = grad.double()
        total = float(g64.norm())
        g64 = g64 * min(1.0, max_norm / (total + 1e-6))
        d = g64 + 1e-4 * p64
        m64 = d.clone() if m64 is None else 0.9 * m64 + d
        p64 = p64 - 0.01 * m64
        gd = dev(grad)
        L.lib.call('simq_clip_sgd_step', L.ptr(pd), L.ptr(gd), L.ptr(md), count, max_norm, 0.01, 0.9, 1e-4,
                   1 if step == 0 else 0, L.ptr(scratch), L.ptr(tn), st)
        assert abs(tn.item() - total) < 1e-6 * total
        assert rel(gd, g64) < 1e-6              # clipped gradient left in place (clip_grad_norm_ semantics)
        assert rel(pd, p64) < 1e-6
        assert rel(md, m64) < 1e-6
        if count < 100000:                      # and against torch's own implementation
            pt.grad = grad.clone()
            torch.nn.utils.clip_grad_norm_([pt], max_norm)
            opt.step()
            assert rel(pd, pt.detach()) < 1e-6
            assert rel(md, opt.state[pt]['momentum_buffer']) < 1e-5


def test_replay_gather(L):
    ring = torch.randn(20, 96, 96, 5)
    idx = torch.tensor([3, 19, 0, 3], dtype=torch.int64)
    out = torch.empty(4, 96, 96, 5, device='cuda')
    ring_d, idx_d = dev(ring), dev(idx)
    L.lib.call('simq_replay_gather', L.ptr(ring_d), 96 * 96 * 5, L.ptr(idx_d), 4, L.ptr(out), L.stream_ptr())
    assert torch.equal(out.cpu(), ring[idx])


def test_error_reporting(L):
    rc = L.lib.c.simq_plan_create(4, 9, ctypes.byref(ctypes.c_void_p()))
    assert rc != 0 and 'num_output_channels' in L.last_error()
    with pytest.raises(L.SimqError):
        L.Plan(0, 1)


BF16_CASES = [
    # B, H, Cin, Cout, k, pad, bias
    (4, 24, 128, 256, 3, 1, False),
    (2, 24, 512, 512, 3, 1, False),
    (3, 24, 64, 64, 3, 1, False),          # M = 1728: ragged tiles
    (2, 24, 512, 128, 1, 0, True),         # head conv1
    (2, 48, 128, 32, 1, 0, True),          # head conv2 (N = 32)
    (2, 24, 256, 512, 1, 0, False),        # downsample
]


@pytest.mark.parametrize('nplanes,tol_fwd', [(2, 5e-5), (1, 3e-2)], ids=['split_bf16x3', 'bf16'])
@pytest.mark.parametrize('case', BF16_CASES, ids=lambda c: 'B%d_H%d_%dto%d_k%d' % (c[0], c[1], c[2], c[3], c[4]))
def test_conv_bf16_matrix_core_paths(L, case, nplanes, tol_fwd):
    """bf16 MFMA kernels (plain bf16 and split-bf16 hi/lo with 3 products) vs an fp64 convolution.
    split-bf16 must be fp32-class (<= 5e-5 of max|y|, typically 3e-6); plain bf16 is bf16-class."""
    B, H, Cin, Cout, k, pad, bias = case
    g = torch.Generator().manual_seed(99 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    xd64 = x.double().requires_grad_(True)
    wd64 = w.double().requires_grad_(True)
    y_ref = F.conv2d(xd64, wd64, None if b is None else b.double(), padding=pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    st = L.stream_ptr()
    xd, wd, dyd = nhwc(x), ohwi(w), nhwc(dy)
    bd = dev(b) if bias else None
    yd = torch.empty(B, H, H, Cout, device='cuda')
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    scratch = torch.empty(2 * (x.numel() + max(w.numel(), dy.numel())) + 64, dtype=torch.int16, device='cuda')
    L.lib.call('simq_conv2d_fwd_bf16', L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(yd), B, H, H, Cin, Cout, k, k, 1, pad, nplanes,
               L.ptr(scratch), L.ptr(stats), st)
    y_nhwc = y_ref.detach().permute(0, 2, 3, 1)
    assert rel(yd, y_nhwc) < tol_fwd
    assert rel(stats[:Cout], yd.double().reshape(-1, Cout).sum(0)) < 1e-5      # statistics of what was written
    dwd = torch.full((Cout, k, k, Cin), 7.0, device='cuda')
    L.lib.call('simq_conv2d_wgrad_bf16', L.ptr(xd), L.ptr(dyd), L.ptr(dwd), B, H, H, Cin, Cout, k, k, 1, pad, nplanes,
               L.ptr(scratch), st)
    assert rel(dwd, wd64.grad.permute(0, 2, 3, 1)) < tol_fwd
    if nplanes == 1:
        # tight form of the bf16 bar: fp64 on the bf16-ROUNDED operands -- the products of bf16 values are exact in fp32, so only the
        # accumulation order is the kernel's own and 2e-5 holds; a dropped tap / K-tile / pixel row fails it by orders of magnitude
        xr, wr, dyr = x.bfloat16().double(), w.bfloat16().double().requires_grad_(True), dy.bfloat16().double()
        yr = F.conv2d(xr, wr, None if b is None else b.double(), padding=pad)
        assert rel(yd, yr.detach().permute(0, 2, 3, 1)) < 2e-5
        yr.backward(dyr)
        assert rel(dwd, wr.grad.permute(0, 2, 3, 1)) < 2e-5


@pytest.mark.parametrize('B,Cin,Cout', [(128, 128, 128), (128, 64, 64), (64, 64, 128)], ids=['layer2_b128', 'layer1_b128', 'l2a_b64'])
def test_conv_bf16_register_staged_wgrad_at_config_batch_sizes(L, B, Cin, Cout):
    """The register-staged bf16 weight-gradient kernel (conv_wgrad_bf16.hip: the 64- and 128-channel layers, which the 256x256
    ping-pong tile does not cover) at BASELINE configs[2..4]'s per-GPU batch sizes, against fp64 on the bf16-rounded operands."""
    H, k = 24, 3
    g = torch.Generator().manual_seed(17 + Cin + Cout + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    dy = torch.randn(B, H, H, Cout, generator=g).cuda()
    scratch = torch.empty(2 * (x.numel() + dy.numel()) + 64, dtype=torch.int16, device='cuda')
    d1 = torch.full((Cout, k, k, Cin), 7.0, device='cuda')
    L.lib.call('simq_conv2d_wgrad_bf16', L.ptr(x), L.ptr(dy), L.ptr(d1), B, H, H, Cin, Cout, k, k, 1, 1, 1, L.ptr(scratch), L.stream_ptr())
    xb, dyb = x.bfloat16().double(), dy.bfloat16().double()
    ref = torch.nn.grad.conv2d_weight(xb.permute(0, 3, 1, 2), (Cout, Cin, k, k), dyb.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(d1).all()
    assert rel(d1, ref) < 2e-5, rel(d1, ref)


@pytest.mark.parametrize('B,Cin,Cout,tile', [(8, 256, 256, (576, 128)), (7, 128, 256, (576, 128)), (8, 256, 512, (288, 256)), (7, 64, 256, (288, 256)),
                                             (7, 64, 64, (144, 64)), (128, 256, 256, (576, 128)), (128, 64, 64, (144, 64)),
                                             (7, 64, 64, (288, 64)), (3, 64, 128, (288, 64)), (128, 64, 64, (288, 64)),
                                             (7, 128, 128, (1288, 128)), (128, 128, 128, (1288, 128)), (5, 256, 256, (1288, 128)), (64, 64, 128, (1288, 128))],
                         ids=['image_tile_l3', 'image_tile_l3a_b7', 'pingpong_l4a', 'pingpong_ragged', 'dma_144x64_layer1',
                              'image_tile_l3_b128', 'dma_144x64_layer1_b128', 'resident_c64_b7', 'resident_c64_to128_b3', 'resident_c64_b128',
                              'half_image_tile_l2_b7', 'half_image_tile_l2_b128', 'half_image_tile_l3_b5', 'half_image_tile_64to128_b64'])
def test_conv_bf16_pingpong_kernels_match_register_staged(L, B, Cin, Cout, tile):
    """The two ping-pong bf16 kernels of the 3x3 layers on the 24x24 maps -- conv_igemm_bf16_img.hip (tile "576x128": one image x 128
    channels per block, halo patch staged once per 32-channel chunk, nine taps read shifted fragments) and conv_igemm_bf16_pp.hip
    (288x256 implicit-GEMM tile, 4-stage LDS-DMA ring) -- against the register-staged kernel on the same bf16 operands: same
    products, fp32 accumulation in a different K order, so outputs agree to fp32 round-off; bias + batch statistics through the
    staged epilogue.  Odd batches: ragged M for the 288-row tile, an odd image count for the image tile.  "144 x 64": the LDS-DMA tile of
    the 64-channel layer1 (conv_igemm_bf16_dma.hip); "288 x 64": conv_igemm_bf16_c64.hip, which large-batch plans use for the
    64-input-channel layers instead (half an image x 64 channels per block, patch and all nine taps of the weights resident in LDS);
    "1288 x 128": the HALF-map form of the image-tile kernel (12 image rows x 128 channels per block, 2 x 4 waves of 144 x 32: round 4, the
    128-channel layers at B = 128 where whole maps give only 128 blocks)."""
    H, k = 24, 3
    g = torch.Generator().manual_seed(11 + Cin + Cout + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    st = L.stream_ptr()
    outs = []
    for t in (tile, (96, 128) if Cout % 128 == 0 else (96, 64)):       # (the second: a register-staged tile)
        y = torch.full((B, H, H, Cout), float('nan'), device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, 1,
                   L.ptr(scratch), L.ptr(stats), st, opts=L.launch_opts(tile=t))
        outs.append((y, stats))
    (y1, s1), (y0, s0) = outs
    assert torch.isfinite(y1).all()
    assert rel(y1, y0) < 5e-6 and rel(s1, s0) < 1e-6
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert rel(y1, ref) < 3e-2                                                   # bf16-class against fp64
    # ... and TIGHT against fp64 on the bf16-rounded operands: bf16 products are exact in fp32, only the accumulation order is the
    # kernel's own -- a kernel that drops or doubles a tap / K-tile / halo row at any batch size fails this by orders of magnitude
    refr = F.conv2d(x.bfloat16().double().permute(0, 3, 1, 2), w.bfloat16().double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
    assert rel(y1, refr) < 2e-5, rel(y1, refr)
    mean64 = refr.reshape(-1, Cout).sum(0)
    assert rel(s1[:Cout], mean64) < 1e-4                                          # batch statistics from the fp32 accumulators


@pytest.mark.parametrize('B,Cin,Cout', [(8, 256, 256), (5, 512, 256), (6, 256, 512), (128, 256, 256), (64, 512, 512), (100, 128, 256), (37, 512, 256)],
                         ids=['l3', 'l4b_ragged', 'l4a', 'l3_b128', 'l4_b64', 'l3a_b100_ragged_splits', 'l4b_b37_odd_images'])
def test_conv_bf16_wgrad_pingpong_matches_register_staged(L, B, Cin, Cout):
    """conv_wgrad_bf16_pp.hip (256x256 tiles per tap, LDS-DMA staged rows read back by ds_read_b64_tr_b16, ping-pong wave groups,
    pixel reduction split over blocks) against the register-staged 128x128 wgrad kernel (SIMQ-internal switch) and fp64: the same
    bf16 products, fp32 accumulation / atomics in a different order."""
    import os
    H, k = 24, 3
    g = torch.Generator().manual_seed(13 + Cin + Cout + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    dy = torch.randn(B, H, H, Cout, generator=g).cuda()
    scratch = torch.empty(2 * (x.numel() + dy.numel()) + 64, dtype=torch.int16, device='cuda')
    st = L.stream_ptr()
    d1 = torch.full((Cout, k, k, Cin), 7.0, device='cuda')
    L.lib.call('simq_conv2d_wgrad_bf16', L.ptr(x), L.ptr(dy), L.ptr(d1), B, H, H, Cin, Cout, k, k, 1, 1, 1, L.ptr(scratch), st)
    # reference: bf16-rounded operands, exact products, fp64 accumulation
    xb, dyb = x.bfloat16().double(), dy.bfloat16().double()
    ref = torch.nn.grad.conv2d_weight(xb.permute(0, 3, 1, 2), (Cout, Cin, k, k), dyb.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(d1).all()
    assert rel(d1, ref) < 2e-5, rel(d1, ref)                                     # fp32 accumulation of exact bf16 products
    # the form the plans use: per-block partial tiles in a slab + a fixed-order reduction launch (image-tile kernel) -- deterministic
    slab = torch.empty(L._c.simq_conv2d_wgrad_bf16_slab_bytes() // 4, device='cuda')
    d2, d3 = torch.full_like(d1, 7.0), torch.full_like(d1, -3.0)
    for d in (d2, d3):
        slab.fill_(float('nan'))
        L.lib.call('simq_conv2d_wgrad_bf16_slab', L.ptr(x), L.ptr(dy), L.ptr(d), B, H, H, Cin, Cout, k, k, 1, 1, 1, L.ptr(scratch), L.ptr(slab), st)
    assert rel(d2, ref) < 2e-5, rel(d2, ref) and rel(d3, ref) < 2e-5
    if B >= 37:                       # (the image-tile kernel takes the launch from two images per block; below that the atomics forms run)
        assert torch.equal(d2, d3)


@pytest.mark.parametrize('B,H,Cin,Cout,k', [(16, 24, 128, 256, 3), (15, 24, 64, 128, 3), (16, 24, 256, 128, 1), (128, 24, 128, 128, 3),
                                            (128, 24, 256, 512, 1)],
                         ids=['256tiles', 'ragged_rows', '1x1', 'layer2_b128', 'downsample_1x1_b128'])
def test_conv_bf16_lds_dma_kernel_matches_register_staged(L, B, H, Cin, Cout, k):
    """The large-tile LDS-DMA kernel (conv_igemm_bf16_dma.hip, 288x128 tile, 8 waves, staged vector epilogue) against
    the register-staged kernel on the same bf16 operands: identical K order (64-channel chunk outer, tap inner) and
    fp32 MFMA accumulation, so outputs agree to fp32 round-off; bias + batch statistics go through both epilogues.
    'ragged_rows': M = 15*576 is not a multiple of 288 (zero-filled DMA rows, masked stores)."""
    g = torch.Generator().manual_seed(7 + Cin + Cout)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    scratch = torch.empty(2 * (x.numel() + w.numel()) + 64, dtype=torch.int16, device='cuda')
    st = L.stream_ptr()
    outs = []
    for tile in ((288, 128), (96, 128)):
        y = torch.full((B, H, H, Cout), float('nan'), device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        L.lib.call('simq_conv2d_fwd_bf16', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, k // 2, 1,
                   L.ptr(scratch), L.ptr(stats), st, opts=L.launch_opts(tile=tile))
        outs.append((y, stats))
    (y_dma, s_dma), (y_reg, s_reg) = outs
    assert torch.isfinite(y_dma).all()
    assert rel(y_dma, y_reg) < 2e-6
    assert rel(s_dma, s_reg) < 1e-6
    # and both are bf16-class against an fp64 convolution of the same inputs
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    assert rel(y_dma, ref) < 2e-2
    # tight: fp64 on the bf16-rounded operands (exact products, only the accumulation order is the kernel's)
    refr = F.conv2d(x.bfloat16().double().permute(0, 3, 1, 2), w.bfloat16().double().permute(0, 3, 1, 2), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    assert rel(y_dma, refr) < 2e-5, rel(y_dma, refr)


@pytest.mark.parametrize('B,H', [(29, 24), (39, 20)], ids=['112_tail_tiles', 'ragged_tail'])
def test_conv_fp32_balanced_last_round(L, B, H):
    """conv_igemm.hip's opt-in balanced last round (simq_launch_opts.tail_split / simq_plan_options.tail_split): the tiles of a partial last round are contracted
    in K-slices by several blocks and finished by igemm_tail_fixup_kernel.  Same output (up to the fp32 summation order of
    the slices) and the same fused bias + batch statistics as the plain launch.  (29, 24): 1392 tiles of 96x64 on 1280
    slots; (39, 20): M = 15600 is not a multiple of 96 -- the 24 tail tiles include the ragged last row tile."""
    Cin, Cout, k = 512, 512, 3
    g = torch.Generator().manual_seed(91 + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    st = L.stream_ptr()
    outs = []
    for on in (0, 1):
        y = torch.full((B, H, H, Cout), float('nan'), device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, L.ptr(stats), st,
                   opts=L.launch_opts(tile=(96, 64), tail_split=on))
        outs.append((y, stats))
    (y0, s0), (y1, s1) = outs
    assert torch.isfinite(y1).all()
    assert rel(y1, y0) < 5e-6 and rel(s1, s0) < 1e-6
    assert not torch.equal(y1, y0)                           # the sliced path really ran (different summation order)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert rel(y1, ref) < 1e-4


@pytest.mark.parametrize('B,Cout,bias', [(1, 64, False), (3, 128, True), (32, 64, False), (29, 64, True)], ids=['b1', 'b3_to128_bias', 'b32', 'b29_bias'])
def test_conv_fp32_image_tile_kernel_matches_implicit_gemm(L, B, Cout, bias):
    """conv_img_f32.hip (three image rows x 64 output channels per block, halo patch in LDS) serves the 3x3 convolutions with 64 input
    channels on the 24x24 maps; the implicit-GEMM kernel (forced 32x32 tile) computes the same exact-fp32 FMA products in another
    order.  Output + fused bias + batch statistics against each other and against the fp64 convolution; borders carry large values
    so that a wrong halo shows."""
    Cin, k, H = 64, 3, 24
    g = torch.Generator().manual_seed(4321 + B + Cout)
    x = torch.randn(B, H, H, Cin, generator=g)
    x[:, 0, :, :] *= 8.0; x[:, -1, :, :] *= 8.0; x[:, :, 0, :] *= 8.0; x[:, :, -1, :] *= 8.0
    x = x.cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() if bias else None
    st = L.stream_ptr()
    outs = []
    for tile in ((0, 0), (32, 32)):
        y = torch.full((B, H, H, Cout), float('nan'), device='cuda')
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
        L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, H, H, Cin, Cout, k, k, 1, 1, L.ptr(stats), st,
                   opts=L.launch_opts(tile=tile))
        outs.append((y, stats))
    (y0, s0), (y1, s1) = outs
    assert torch.isfinite(y0).all()
    assert rel(y0, y1) < 5e-6 and rel(s0, s1) < 1e-6
    assert not torch.equal(y0, y1)                           # the image-tile kernel really ran (different summation order)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double() if bias else None, padding=1).permute(0, 2, 3, 1)
    assert rel(y0, ref) < 2e-6
    s_ref = ref.reshape(-1, Cout)
    assert rel(s0[:Cout], s_ref.sum(0)) < 1e-5 and rel(s0[Cout:], (s_ref * s_ref).sum(0)) < 1e-5


@pytest.mark.parametrize('B,H,Cin,Cout', [(5, 24, 512, 512), (3, 24, 256, 512), (4, 12, 128, 256), (2, 8, 64, 64), (7, 24, 128, 128)],
                         ids=['l4', 'l4a', 'l3a_small_map', 'narrow', 'l2'])
def test_conv_winograd_forward_matches_direct(L, B, H, Cin, Cout):
    """conv_winograd.hip (input transform, 16 batched transform-domain GEMMs, output transform + epilogue) against the
    implicit-GEMM kernel and an fp64 convolution: outputs within 1e-5 of the output range (the transforms only add and
    halve, measured ~7e-7), fused bias + batch statistics identical to round-off."""
    g = torch.Generator().manual_seed(17 + Cin + Cout + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (Cin * 9) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    T = B * (H // 2) ** 2
    scratch = torch.empty(16 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda')
    st = L.stream_ptr()
    y0, y1 = torch.empty(B, H, H, Cout, device='cuda'), torch.full((B, H, H, Cout), float('nan'), device='cuda')
    s0, s1 = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda'), torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y0), B, H, H, Cin, Cout, 3, 3, 1, 1, L.ptr(s0), st)
    L.lib.call('simq_conv2d_fwd_winograd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y1), B, H, H, Cin, Cout, L.ptr(s1), L.ptr(scratch), st)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(y1).all()
    assert rel(y1, ref) < 1e-5 and rel(y1, y0) < 1e-5
    assert rel(s1, s0) < 1e-6
    sref = torch.cat([ref.reshape(-1, Cout).sum(0), (ref * ref).reshape(-1, Cout).sum(0)])
    assert rel(s1, sref) < 1e-5


@pytest.mark.parametrize('f4,B,H,Cin,Cout', [(False, 5, 24, 512, 512), (True, 5, 24, 512, 512), (True, 32, 24, 512, 512), (True, 3, 24, 256, 512),
                                             (True, 7, 12, 256, 256), (False, 29, 24, 256, 256), (True, 1, 24, 128, 128)],
                         ids=['f2_l4', 'f4_l4', 'f4_l4_b32', 'f4_l4a', 'f4_small_map_odd_tiles', 'f2_l3_b29', 'f4_one_image'])
def test_winograd_gemm_plane_per_xcd_walk_is_bit_identical(L, f4, B, H, Cin, Cout):
    """conv_igemm.hip's batched transform-domain GEMM hands whole planes to one XCD (16 planes: two per XCD; 36: four per XCD and the
    last four shared by two XCDs each, a contiguous half of the row blocks per XCD).  It is a permutation of (plane, tile) over the
    launch's blocks: every block must be covered exactly once -- the output equals the launch-order walk (simq_launch_opts.plane_xcd = 0) BIT FOR
    BIT, with a NaN-filled destination to catch an uncovered tile.  Shapes whose tile count does not divide (odd row-block counts)
    fall back to the launch-order walk inside the launcher and are covered by the same equality."""
    g = torch.Generator().manual_seed(23 + Cin + Cout + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (Cin * 9) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    nb, T = (36, B * (H // 4) ** 2) if f4 else (16, B * (H // 2) ** 2)
    scratch = torch.empty(nb * Cout * Cin + nb * T * (Cin + Cout) + 64, device='cuda')
    name = 'simq_conv2d_fwd_winograd4' if f4 else 'simq_conv2d_fwd_winograd'
    ys = []
    for on in (0, 1):
        y = torch.full((B, H, H, Cout), float('nan'), device='cuda')
        scratch.fill_(float('nan'))
        L.lib.call(name, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, H, H, Cin, Cout, None, L.ptr(scratch), L.stream_ptr(),
                   opts=L.launch_opts(plane_xcd=on))
        torch.cuda.synchronize()
        ys.append(y)
    assert torch.isfinite(ys[1]).all()
    assert torch.equal(ys[0], ys[1])


@pytest.mark.parametrize('B,H,Cin,Cout', [(5, 24, 512, 512), (3, 24, 256, 512), (4, 24, 128, 256), (6, 12, 256, 256), (29, 24, 256, 256),
                                          (9, 24, 128, 128)],
                         ids=['l4', 'l4a', 'l3a', 'l3_small_map', 'l3_b29', 'l2'])
def test_conv_winograd4_forward_matches_direct(L, B, H, Cin, Cout):
    """The F(4x4,3x3) form of the no-grad forwards (conv_winograd.hip: 6x6 input transform at stride 4, 36 batched transform-domain
    GEMMs, 4x4 output transform + forward epilogue; interpolation points {0, 1, -1, 1/2, -2, inf}) against the implicit-GEMM kernel and
    an fp64 convolution: its larger transform coefficients cost ~6x the round-off of F(2x2,3x3) -- measured / simulated 3-4e-6 of the
    output range -- held to 2e-5 per layer (the Q-map bar after the eight wide layers is 1e-4); fused bias + batch statistics agree
    with the direct kernel to the same order."""
    g = torch.Generator().manual_seed(19 + Cin + Cout + B)
    x = torch.relu(torch.randn(B, H, H, Cin, generator=g)).cuda()          # post-ReLU activations, as in the network
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (Cout * 9)) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    T4 = B * (H // 4) ** 2
    scratch = torch.empty(36 * Cout * Cin + 36 * T4 * (Cin + Cout), device='cuda')
    st = L.stream_ptr()
    y0, y1 = torch.empty(B, H, H, Cout, device='cuda'), torch.full((B, H, H, Cout), float('nan'), device='cuda')
    s0, s1 = torch.zeros(2 * Cout, dtype=torch.float64, device='cuda'), torch.zeros(2 * Cout, dtype=torch.float64, device='cuda')
    L.lib.call('simq_conv2d_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y0), B, H, H, Cin, Cout, 3, 3, 1, 1, L.ptr(s0), st)
    L.lib.call('simq_conv2d_fwd_winograd4', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y1), B, H, H, Cin, Cout, L.ptr(s1), L.ptr(scratch), st)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(y1).all()
    print('\nF(4x4,3x3) forward %dx%d B=%d: error vs fp64 %.3g (direct kernel %.3g)' % (Cin, Cout, B, rel(y1, ref), rel(y0, ref)))
    assert rel(y1, ref) < 2e-5 and rel(y1, y0) < 2e-5
    assert rel(s1, s0) < 1e-5
    sref = torch.cat([ref.reshape(-1, Cout).sum(0), (ref * ref).reshape(-1, Cout).sum(0)])
    assert rel(s1, sref) < 1e-5


@pytest.mark.parametrize('B,H,Cin,Cout', [(5, 24, 512, 512), (3, 24, 256, 512), (6, 12, 256, 256), (8, 24, 512, 512), (4, 24, 256, 256),
                                          (8, 24, 128, 128), (5, 24, 128, 128)],
                         ids=['l4', 'l4a', 'l3_small_map', 'l4_f4', 'l3_f4', 'l2_f4', 'l2'])
def test_conv_winograd_wgrad_matches_direct(L, B, H, Cin, Cout):
    """Transform-domain weight gradient (dy / x transforms, batched contractions over the tiles, G^T dU G) against the
    direct wgrad kernel and fp64.  Tile counts that allow it ('*_f4': batch * 36 a multiple of 16) take the F(4x4,3x3) form,
    whose larger coefficients cost ~10x the round-off (measured 0.6-3e-5 of the gradient's range) -- held to the 1e-4
    per-kernel bar (a weight gradient is a leaf: the error does not propagate); the others run F(2x2,3x3) (GEMM form, or the pixel-split form when the tile count is not a multiple of 16)."""
    g = torch.Generator().manual_seed(29 + Cin + Cout + B)
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    dy = torch.randn(B, H, H, Cout, generator=g).cuda()
    T = B * (H // 2) ** 2
    scratch = torch.empty(36 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda')
    st = L.stream_ptr()
    d0, d1 = torch.empty(Cout, 3, 3, Cin, device='cuda'), torch.full((Cout, 3, 3, Cin), float('nan'), device='cuda')
    L.lib.call('simq_conv2d_wgrad', L.ptr(x), L.ptr(dy), L.ptr(d0), B, H, H, Cin, Cout, 3, 3, 1, 1, st)
    L.lib.call('simq_conv2d_wgrad_winograd', L.ptr(x), L.ptr(dy), L.ptr(d1), B, H, H, Cin, Cout, L.ptr(scratch), st)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(),
                                      padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(d1).all()
    assert rel(d1, ref) < (1e-4 if B % 4 == 0 else 1e-5) and rel(d0, ref) < 1e-5


@pytest.mark.parametrize('B,Cin,Cout,splits', [(32, 256, 256, 0), (32, 256, 256, 2), (32, 256, 256, 4), (32, 128, 256, 4), (16, 256, 512, 2),
                                                (8, 256, 256, 4), (32, 512, 512, 0), (12, 256, 256, 0)],
                         ids=['l3_auto', 'l3_s2', 'l3_s4', 'l3a_s4', 'l4a_s2', 'l3_b8_s4_no_room_or_short', 'l4_auto_stays_whole', 'b12_tiles_not_divisible'])
def test_conv_winograd_wgrad_ksplit(L, B, Cin, Cout, splits):
    """K-split of the F(4x4,3x3) weight-gradient GEMMs (conv_winograd.hip: S chunks of the tile range as 36 S planes, summed in chunk order
    by wino4_dw_kernel): forced and shape-chosen splits against fp64 and against the unsplit form -- same 1e-4 per-kernel bar as the
    unsplit F(4x4,3x3) gradient (the split shortens every accumulation chain), NaN-filled destination and scratch, and a second call that
    must reproduce the first bit for bit (no atomics)."""
    g = torch.Generator().manual_seed(31 + Cin + Cout + B)
    H = 24
    x = torch.randn(B, H, H, Cin, generator=g).cuda()
    dy = torch.randn(B, H, H, Cout, generator=g).cuda()
    T = B * (H // 2) ** 2
    scratch = torch.empty(36 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda')
    st = L.stream_ptr()
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(),
                                      padding=1).permute(0, 2, 3, 1)
    out = {}
    for s in (1, splits, splits):
        d = torch.full((Cout, 3, 3, Cin), float('nan'), device='cuda')
        scratch.fill_(float('nan'))
        L.lib.call('simq_conv2d_wgrad_winograd', L.ptr(x), L.ptr(dy), L.ptr(d), B, H, H, Cin, Cout, L.ptr(scratch), st,
                   opts=L.launch_opts(wgrad_ksplit=s))
        torch.cuda.synchronize()
        assert torch.isfinite(d).all()
        assert rel(d, ref) < 1e-4
        out.setdefault(s, []).append(d)
    assert torch.equal(out[splits][0], out[splits][1])
    assert rel(out[splits][0], out[1][0]) < 6e-5               # (two summation orders of a gradient whose own error is 0.6-3e-5)


@pytest.mark.parametrize('M,N,K,P', [(1152, 512, 512, 36), (1044, 256, 128, 36), (4608, 128, 128, 16), (512, 512, 1152, 36), (200, 64, 48, 5)],
                         ids=['f4_l4_b32', 'f4_b29_ragged_rows', 'f2_l2_b32', 'wgrad_l4', 'small_odd_planes'])
def test_gemm_f32_batched_is_an_exact_fp32_contraction(L, M, N, K, P):
    """simq_gemm_f32_batched -- the transform-domain contraction of the Winograd layers, the dominant kernel of the fp32 step, on its own:
    P independent y_g = x_g * w_g^T against fp64 at fp32 round-off (v_mfma_f32_16x16x4_f32 chains: 1e-6 of the range), a NaN-filled
    destination (every tile written, rows past M never), and the two block -> (plane, tile) walks bit-identical."""
    g = torch.Generator().manual_seed(5 + M + N + K)
    x = torch.randn(P, M, K, generator=g).cuda(); w = torch.randn(P, N, K, generator=g).cuda()
    ref = torch.bmm(x.double(), w.double().transpose(1, 2))
    ys = []
    for on in (1, 0):
        y = torch.full((P, M, N), float('nan'), device='cuda')
        L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(plane_xcd=on))
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        ys.append(y)
    assert float((ys[0].double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert torch.equal(ys[0], ys[1])
    with pytest.raises(Exception, match='not supported'):
        L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(ys[0]), M, N - 1, K, P, L.stream_ptr())


@pytest.mark.parametrize('M,N,K,P,tile', [(1152, 512, 512, 36, None), (1152, 512, 512, 36, (128, 128)), (1044, 256, 128, 36, None), (4608, 128, 128, 16, None),
                                          (512, 512, 1152, 36, None), (256, 256, 576, 72, None), (200, 128, 48, 5, None), (70, 256, 512, 9, (128, 256))],
                         ids=['f4_l4_b32_wide_tile', 'f4_l4_b32_128x128', 'f4_b29_ragged_rows', 'f2_l2_b32', 'wgrad_l4', 'wgrad_l3_ksplit', 'small_odd_planes', 'one_ragged_tile_wide'])
def test_gemm_split3_reproduces_the_fp32_contraction_at_fp32_roundoff(L, M, N, K, P, tile):
    """Round 6: simq_gemm_f32_batched with gemm_split = 1 (gemm_split3.hip) -- the same contraction on the bf16 matrix cores: both fp32 operands
    split EXACTLY into three bf16 pieces while they are staged, the six partial products down to 2^-24 of each product, fp32 accumulators,
    fp32 output.  Held to the bars of the fp32-MFMA form (2e-6 of the range against fp64) AND to that form itself: its rms error against
    fp64 may not exceed 1.1 x the fp32-MFMA kernel's on the same operands (measured 0.85 x: the matrix core rounds once per 16 products
    instead of once per product); NaN-filled destination, rows past M never written, both block walks bit-identical; operands with a 2^40
    dynamic range (every piece of the split in play) and exactly representable integers (result EXACT: no piece is lost)."""
    g = torch.Generator().manual_seed(7 + M + N + K)
    x = torch.randn(P, M, K, generator=g).cuda(); w = torch.randn(P, N, K, generator=g).cuda()
    ref = torch.bmm(x.double(), w.double().transpose(1, 2))
    y0 = torch.empty(P, M, N, device='cuda')
    L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y0), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(gemm_split=0))
    ys = []
    for on in (1, 0):
        y = torch.full((P * M * N + 3 * N,), float('nan'), device='cuda')          # (three guard rows behind the last plane: rows past M)
        L.lib.call('simq_launch_counts_reset')
        L.lib.call('simq_gemm_f32_batched', L.ptr(x), L.ptr(w), L.ptr(y), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(tile=tile, plane_xcd=on, gemm_split=1))
        torch.cuda.synchronize()
        assert L.launch_counts().get('gemm_split3_batched', 0) == 1
        assert torch.isnan(y[P * M * N:]).all(), 'rows past M of the last plane were written'
        y = y[:P * M * N].view(P, M, N)
        assert torch.isfinite(y).all()
        ys.append(y)
    assert torch.equal(ys[0], ys[1])
    err = lambda t: float((t.double() - ref).pow(2).mean().sqrt())
    assert float((ys[0].double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert err(ys[0]) <= 1.1 * err(y0), (err(ys[0]), err(y0))
    # wide dynamic range: magnitudes over 2^40 in both operands
    xs = x * torch.exp2(torch.randint(-20, 20, x.shape, generator=g).float().cuda()); ws = w * torch.exp2(torch.randint(-20, 20, w.shape, generator=g).float().cuda())
    refs = torch.bmm(xs.double(), ws.double().transpose(1, 2))
    ya, yb = torch.empty(P, M, N, device='cuda'), torch.empty(P, M, N, device='cuda')
    L.lib.call('simq_gemm_f32_batched', L.ptr(xs), L.ptr(ws), L.ptr(ya), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(gemm_split=0))
    L.lib.call('simq_gemm_f32_batched', L.ptr(xs), L.ptr(ws), L.ptr(yb), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(tile=tile, gemm_split=1))
    ea, eb = float((ya.double() - refs).pow(2).mean().sqrt()), float((yb.double() - refs).pow(2).mean().sqrt())
    assert eb <= 2.0 * ea, (eb, ea)
    # integers below 2^11 (11 significant bits: two pieces of the split), products and sums exact in fp32: the result must be EXACT
    xi = torch.randint(-2047, 2048, (P, M, K), generator=g).float().cuda(); wi = torch.randint(-3, 4, (P, N, K), generator=g).float().cuda()
    yi = torch.empty(P, M, N, device='cuda')
    L.lib.call('simq_gemm_f32_batched', L.ptr(xi), L.ptr(wi), L.ptr(yi), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(tile=tile, gemm_split=1))
    assert torch.equal(yi.double(), torch.bmm(xi.double(), wi.double().transpose(1, 2)))
    # ... and 24-bit integers (all three pieces of the split carry bits) against small ones: the sums need more than 24 bits, so the result is
    # rounded -- no worse than the fp32-MFMA form rounds it
    xi = torch.randint(-(1 << 24) + 1, 1 << 24, (P, M, K), generator=g).float().cuda()
    refi = torch.bmm(xi.double(), wi.double().transpose(1, 2))
    L.lib.call('simq_gemm_f32_batched', L.ptr(xi), L.ptr(wi), L.ptr(yi), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(tile=tile, gemm_split=1))
    e1 = float((yi.double() - refi).pow(2).mean().sqrt())
    L.lib.call('simq_gemm_f32_batched', L.ptr(xi), L.ptr(wi), L.ptr(yi), M, N, K, P, L.stream_ptr(), opts=L.launch_opts(gemm_split=0))
    e0 = float((yi.double() - refi).pow(2).mean().sqrt())
    assert e1 <= 1.1 * e0 + 1e-30, (e1, e0)


def test_conv_winograd_rejects_unsupported_geometry(L):
    """Odd map sizes / channel counts the transform kernels cannot tile are refused with a message, not mis-computed."""
    x = torch.zeros(1, 23, 23, 64, device='cuda'); w = torch.zeros(64, 3, 3, 64, device='cuda'); y = torch.zeros(1, 23, 23, 64, device='cuda')
    scratch = torch.zeros(1 << 20, device='cuda')
    with pytest.raises(Exception, match='geometry not supported'):
        L.lib.call('simq_conv2d_fwd_winograd', L.ptr(x), L.ptr(w), None, L.ptr(y), 1, 23, 23, 64, 64, None, L.ptr(scratch), L.stream_ptr())
    with pytest.raises(Exception, match='geometry not supported'):
        L.lib.call('simq_conv2d_wgrad_winograd', L.ptr(x), L.ptr(y), L.ptr(w), 1, 24, 24, 64, 64, L.ptr(scratch), L.stream_ptr())


@pytest.mark.parametrize('B,C', [(3, 4), (2, 5), (2, 3), (1, 9), (5, 1), (32, 4), (2, 7), (3, 6)], ids=['cin4', 'cin5', 'cin3', 'cin9', 'cin1', 'cin4_b32', 'cin7', 'cin6'])
def test_stem_conv_f32_matches_an_fp64_convolution(L, B, C):
    """stem_conv_f32.hip (reference resnet.py:94: conv 7x7 s2 p3 -> 64, the form fp32 / split-bf16 plans run): 16-byte runs of the NHWC
    input straight into v_mfma_f32_16x16x4_f32 (the K index of the MFMA permuted consistently on both operands), left / right / top / bottom
    padding by masks, weights from the OHWI parameters.  Exact fp32 FMA chains: the fp32 bar (1e-4; measured ~1e-6) against an fp64
    convolution; batch statistics 1e-5.  Large values on every image edge so that a wrong padding mask cannot hide."""
    g = torch.Generator().manual_seed(43 + 7 * C + B)
    x = torch.randn(B, 96, 96, C, generator=g)
    x[:, 0, :, :] += 3.0; x[:, -1, :, :] -= 3.0; x[:, :, 0, :] += 5.0; x[:, :, -1, :] -= 5.0
    w = torch.randn(64, 7, 7, C, generator=g) * (2.0 / (49 * C)) ** 0.5
    xd, wd = x.cuda(), w.cuda()
    y = torch.full((B, 48, 48, 64), float('nan'), device='cuda')
    stats = torch.zeros(128, dtype=torch.float64, device='cuda')
    L.lib.call('simq_conv2d_fwd_stem_f32', L.ptr(xd), L.ptr(wd), L.ptr(y), B, 96, 96, C, L.ptr(stats), L.stream_ptr())
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, stride=2, padding=3).permute(0, 2, 3, 1)
    got = y.double().cpu()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print('\nstem fp32 C=%d B=%d: max error vs fp64 %.3g of the output range' % (C, B, err))
    assert err < 1e-5
    sref = torch.cat([ref.reshape(-1, 64).sum(0), (ref * ref).reshape(-1, 64).sum(0)])
    assert rel(stats.cpu(), sref) < 1e-5
    # no statistics requested (eval mode): same outputs
    y2 = torch.empty_like(y)
    L.lib.call('simq_conv2d_fwd_stem_f32', L.ptr(xd), L.ptr(wd), L.ptr(y2), B, 96, 96, C, None, L.stream_ptr())
    assert torch.equal(y, y2)


def test_stem_kernels_refuse_ten_input_channels(L):
    """Cin = 10 (the reference's widest intention-channel variant): 7 * Cin = 70 exceeds the 64-float filter row both dedicated stem
    kernels are built around -- they refuse it with a message, and plans of that width run the generic implicit GEMM / fp32 weight
    gradient instead (forward_impl: stem_conv_f32_eligible / stem_conv_bf16_eligible; CONV_CASES above cover that path at Cin = 10)."""
    x = torch.zeros(1, 96, 96, 10, device='cuda'); w = torch.zeros(64, 7, 7, 10, device='cuda'); y = torch.zeros(1, 48, 48, 64, device='cuda')
    with pytest.raises(Exception, match='geometry not supported'):
        L.lib.call('simq_conv2d_fwd_stem_f32', L.ptr(x), L.ptr(w), L.ptr(y), 1, 96, 96, 10, None, L.stream_ptr())
    y16 = torch.zeros(1, 48, 48, 64, dtype=torch.bfloat16, device='cuda'); scratch = torch.empty(58368, dtype=torch.uint8, device='cuda')
    with pytest.raises(Exception, match='geometry not supported'):
        L.lib.call('simq_conv2d_fwd_stem_bf16', L.ptr(x), L.ptr(w), L.ptr(y16), 1, 96, 96, 10, None, L.ptr(scratch), L.stream_ptr())
    from simq import _lib
    net10 = __import__('simq').FCN(10, 1, precision='bf16')
    xx = torch.randn(2, 96, 96, 10, device='cuda')
    net10.train()
    _lib.lib.call('simq_launch_counts_reset')
    net10._forward_raw(xx, _lib.MODE_TRAIN)
    torch.cuda.synchronize()
    ran = _lib.launch_counts()
    assert ran.get('stem_conv_bf16', 0) == 0 and ran.get('stem_conv_f32', 0) == 0 and ran.get('igemm_f32_gather', 0) == 1, ran


@pytest.mark.parametrize('B,C', [(3, 4), (2, 5), (2, 3), (1, 9), (5, 1), (3, 6), (2, 7)], ids=['cin4', 'cin5', 'cin3', 'cin9', 'cin1', 'cin6', 'cin7'])
def test_stem_conv_bf16_matches_a_bf16_operand_convolution(L, B, C):
    """stem_conv_bf16.hip (reference resnet.py:94: conv 7x7 s2 p3 -> 64, the form plain-bf16 plans run): fragments gathered straight from
    the fp32 NHWC input (7*C contiguous floats per filter row, left / right / top / bottom padding by masks), bf16 operands, fp32
    accumulation.  Against an fp64 convolution of the bf16-ROUNDED input and weights the only error left is the accumulation order and
    the final bf16 rounding of the stored output (2^-9 relative); the batch statistics come from the unrounded accumulators.  The
    input carries a large value at every image corner and edge so that a wrong padding mask cannot hide."""
    g = torch.Generator().manual_seed(41 + 7 * C + B)
    x = torch.randn(B, 96, 96, C, generator=g)
    x[:, 0, :, :] += 3.0; x[:, -1, :, :] -= 3.0; x[:, :, 0, :] += 5.0; x[:, :, -1, :] -= 5.0
    w = torch.randn(64, 7, 7, C, generator=g) * (2.0 / (49 * C)) ** 0.5
    xd, wd = x.cuda(), w.cuda()
    y = torch.full((B, 48, 48, 64), float('nan'), dtype=torch.bfloat16, device='cuda')
    stats = torch.zeros(128, dtype=torch.float64, device='cuda')
    scratch = torch.empty(58368, dtype=torch.uint8, device='cuda')
    L.lib.call('simq_conv2d_fwd_stem_bf16', L.ptr(xd), L.ptr(wd), L.ptr(y), B, 96, 96, C, L.ptr(stats), L.ptr(scratch), L.stream_ptr())
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), None, stride=2, padding=3).permute(0, 2, 3, 1)
    got = y.double().cpu()
    assert torch.isfinite(got).all()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max()) / scale
    print('\nstem bf16 C=%d B=%d: max error vs fp64 conv of the rounded operands %.3g of the output range' % (C, B, err))
    assert err < 2.0 ** -8                                                   # one bf16 rounding of the output (+ fp32 accumulation)
    sref = torch.cat([ref.reshape(-1, 64).sum(0), (ref * ref).reshape(-1, 64).sum(0)])
    assert rel(stats.cpu(), sref) < 1e-5                                     # statistics: from the fp32 accumulators


@pytest.mark.parametrize('B,C', [(3, 4), (2, 5), (1, 9), (5, 1), (19, 5), (2, 3), (3, 6), (2, 7)], ids=['cin4', 'cin5', 'cin9', 'cin1', 'cin5_b19', 'cin3', 'cin6', 'cin7'])
def test_stem_wgrad_bf16_matches_a_bf16_operand_weight_gradient(L, B, C):
    """Weight gradient of the first convolution on the bf16 matrix cores (stem_conv_bf16.hip: dy plane and the gathered row fragments
    of x staged TRANSPOSED in LDS, contraction over pixels, per-block slabs added in a fixed order): against the fp64 weight gradient
    of the bf16-ROUNDED operands only the fp32 accumulation order is left (1e-5 of the gradient's range); two runs are bit-identical
    (no atomics).  Large values on the image borders expose a wrong padding mask."""
    g = torch.Generator().manual_seed(43 + 7 * C + B)
    x = torch.randn(B, 96, 96, C, generator=g)
    x[:, 0, :, :] += 3.0; x[:, -1, :, :] -= 3.0; x[:, :, 0, :] += 5.0; x[:, :, -1, :] -= 5.0
    dy = torch.randn(B, 48, 48, 64, generator=g).bfloat16()
    xd, dyd = x.cuda(), dy.cuda()
    dw = torch.full((64, 7, 7, C), float('nan'), device='cuda')
    scratch = torch.empty(512 * 64 * 49 * C, device='cuda')
    call = lambda out: L.lib.call('simq_conv2d_wgrad_stem_bf16', L.ptr(xd), L.ptr(dyd), L.ptr(out), B, 96, 96, C, L.ptr(scratch), L.stream_ptr())
    call(dw)
    dw2 = torch.empty_like(dw)
    call(dw2)
    assert torch.equal(dw, dw2)
    xr = x.bfloat16().double().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(64, C, 7, 7, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xr, w, None, stride=2, padding=3)
    y.backward(dy.double().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1)
    got = dw.double().cpu()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print('\nstem wgrad bf16 C=%d B=%d: max error vs fp64 gradient of the rounded operands %.3g of the range' % (C, B, err))
    assert err < 1e-4


_F32PP_CHILD = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'spatial-intention-maps_amd')]
import numpy as np, torch
from simq import _lib as L
assert L.lib.build_flags == 1, 'expected the ablation build'
out = {}
for f4, (B, Cin, Cout) in ((False, (8, 512, 512)), (True, (16, 256, 512))):
    g = torch.Generator().manual_seed(5 + Cin)
    x = torch.randn(B, 24, 24, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda()
    y = torch.empty(B, 24, 24, Cout, device='cuda')
    T, nb = (B * 36, 36) if f4 else (B * 144, 16)
    scratch = torch.empty(nb * Cout * Cin + nb * T * (Cin + Cout) + 64, device='cuda')
    L.lib.call('simq_conv2d_fwd_winograd4' if f4 else 'simq_conv2d_fwd_winograd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, 24, 24, Cin, Cout,
               None, L.ptr(scratch), L.stream_ptr())
    out['f4' if f4 else 'f2'] = y.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_f32_pingpong_gemm_serves_winograd_layers(L, tmp_path):
    """gemm_f32_pp.hip (LDS-DMA ping-pong form of the batched transform-domain GEMM; measured step-neutral, so it is compiled into the
    ablation build libsimq_ablate.so only, where SIMQ_F32_PP=2 routes every eligible contraction through it): a child process on that
    build must reproduce the product library's Winograd convolutions BIT FOR BIT (v_mfma_f32_16x16x4_f32 is an exact FMA chain and
    the K order per accumulator is the same).  Also pins the split itself: the product reports build flags 0 and ignores SIMQ_*."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    abl = os.path.join(root, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so')
    assert L.lib.build_flags == 0
    assert os.path.exists(abl), 'libsimq_ablate.so missing: __graft_entry__.build() / `make -C spatial-intention-maps_amd/csrc ablate` builds it'
    script = tmp_path / 'child.py'
    script.write_text(_F32PP_CHILD % {'root': root})
    path = str(tmp_path / 'pp.npz')
    r = subprocess.run([sys.executable, str(script), path], env=dict(os.environ, SIMQ_LIBRARY=abl, SIMQ_F32_PP='2'), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    got = np.load(path)
    for f4, (B, Cin, Cout) in ((False, (8, 512, 512)), (True, (16, 256, 512))):
        g = torch.Generator().manual_seed(5 + Cin)
        x = torch.randn(B, 24, 24, Cin, generator=g).cuda()
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda()
        y = torch.empty(B, 24, 24, Cout, device='cuda')
        T, nb = (B * 36, 36) if f4 else (B * 144, 16)
        scratch = torch.empty(nb * Cout * Cin + nb * T * (Cin + Cout) + 64, device='cuda')
        L.lib.call('simq_conv2d_fwd_winograd4' if f4 else 'simq_conv2d_fwd_winograd', L.ptr(x), L.ptr(w), None, L.ptr(y), B, 24, 24, Cin, Cout,
                   None, L.ptr(scratch), L.stream_ptr())
        assert np.array_equal(y.cpu().numpy(), got['f4' if f4 else 'f2'])
