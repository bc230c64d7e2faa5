"""GPU: the train-mode "conv -> BatchNorm -> ReLU -> conv" chain with the BatchNorm + ReLU applied inside the SECOND convolution's
operand staging (simq_plan_options.fuse_bn1_apply; reference resnet.py:34-40, networks.py:18-20), kernel by kernel through the C-ABI
and at network level.

The fused consumers compute conv(relu(fma(y, scale, shift))) -- the same fma / max sequence bn_apply performs -- so against an
fp64 convolution of that activation they are held to the per-kernel bars of tests/test_gpu_ops.py (1e-5 direct / F(2x2,3x3), 2e-5
F(4x4,3x3) forward, 1e-4 F(4x4,3x3) weight gradient).  Shifts are positive on average and borders carry large values: a consumer that
applied the BatchNorm to the zero PADDING (relu(shift) != 0) or dropped a halo row fails by orders of magnitude.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    from simq import _lib
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return _lib


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _operands(B, H, Cin, Cout, k, seed):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(B, H, H, Cin, generator=g)
    y[:, 0, :, :] *= 6.0; y[:, -1, :, :] *= 6.0; y[:, :, 0, :] *= 6.0; y[:, :, -1, :] *= 6.0
    sc = 0.5 + torch.rand(Cin, generator=g)
    sh = 0.3 + 0.5 * torch.randn(Cin, generator=g)
    w = torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5
    a = torch.relu(torch.addcmul(sh.double(), y.double(), sc.double()))          # the activation the fused consumers never store
    return y.cuda(), sc.cuda(), sh.cuda(), w.cuda(), a


FWD_CASES = [
    # B, H, Cin, Cout, k, pad, form, bias
    (2, 24, 64, 64, 3, 1, 0, False),       # image-tile kernel (conv_img_f32.hip): BatchNorm pass over the halo patch in LDS
    (3, 24, 64, 128, 3, 1, 0, True),       # ... two N tiles
    (66, 24, 64, 64, 3, 1, 0, False),      # > 512 image-tile blocks: implicit-GEMM vector loader
    (5, 24, 128, 32, 1, 0, 0, True),       # the head's conv2 (1x1, N = 32 tiles)
    (2, 24, 512, 128, 1, 0, 0, True),      # 1x1, long K
    (3, 12, 256, 64, 3, 1, 0, False),      # other map size through the implicit GEMM
    (7, 24, 128, 128, 3, 1, 1, False),     # F(2x2,3x3) input transform
    (3, 24, 256, 512, 3, 1, 1, True),
    (9, 24, 128, 128, 3, 1, 2, False),     # F(4x4,3x3) input transform
    (5, 24, 512, 512, 3, 1, 2, True),
]


@pytest.mark.parametrize('case', FWD_CASES, ids=lambda c: 'B%d_H%d_%dto%d_k%d_form%d' % (c[0], c[1], c[2], c[3], c[4], c[6]))
def test_conv_with_batchnorm_relu_on_load_matches_fp64(L, case):
    B, H, Cin, Cout, k, pad, form, bias = case
    y, sc, sh, w, a = _operands(B, H, Cin, Cout, k, 700 + Cin + Cout + B + form)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(5)).cuda() if bias else None
    planes = {0: 0, 1: 16, 2: 36}[form]
    T = B * (H // 2) ** 2
    scratch = torch.empty(max(1, planes * Cout * Cin + 16 * T * (Cin + Cout)), device='cuda') if form else None
    out = torch.full((B, H, H, Cout), float('nan'), device='cuda')
    L.lib.call('simq_conv2d_fwd_bnrelu_in', L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(w), L.ptr(b), L.ptr(out), B, H, H, Cin, Cout, k, k, 1, pad,
               form, L.ptr(scratch), L.stream_ptr())
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), b.double().cpu() if bias else None, padding=pad).permute(0, 2, 3, 1)
    assert torch.isfinite(out).all()
    err = rel(out, ref)
    print('\nconv(relu(bn(y))) %d->%d k%d B=%d form %d: %.3g vs fp64' % (Cin, Cout, k, B, form, err))
    assert err < (2e-5 if form == 2 else 1e-5)
    # ... and it is the unfused pair of launches' result to round-off: the activation materialised by torch, then the plain kernel
    a32 = torch.relu(torch.addcmul(sh, y, sc))
    plain = torch.empty_like(out)
    if form == 0:
        L.lib.call('simq_conv2d_fwd', L.ptr(a32), L.ptr(w), L.ptr(b), L.ptr(plain), B, H, H, Cin, Cout, k, k, 1, pad, None, L.stream_ptr())
    elif form == 1:
        L.lib.call('simq_conv2d_fwd_winograd', L.ptr(a32), L.ptr(w), L.ptr(b), L.ptr(plain), B, H, H, Cin, Cout, None, L.ptr(scratch), L.stream_ptr())
    else:
        L.lib.call('simq_conv2d_fwd_winograd4', L.ptr(a32), L.ptr(w), L.ptr(b), L.ptr(plain), B, H, H, Cin, Cout, None, L.ptr(scratch), L.stream_ptr())
    assert rel(out, plain) < 2e-6          # (torch's addcmul may or may not contract to an fma: the same values to an ulp)


WGRAD_CASES = [
    # B, H, Cin, Cout, k, pad, form
    (2, 24, 64, 64, 3, 1, 0),              # layer1 (direct, 64x64 tiles)
    (5, 24, 128, 128, 3, 1, 0),            # layer2 (direct)
    (3, 24, 128, 32, 1, 0, 0),             # head conv2 (1x1, Cout = 32)
    (3, 24, 64, 128, 3, 1, 0),             # 128x64 tiles
    (4, 24, 128, 256, 3, 1, 1),            # transform domain, F(4x4,3x3) (B * 36 % 16 == 0)
    (8, 24, 512, 512, 3, 1, 1),
    (3, 24, 256, 256, 3, 1, 1),            # transform domain, F(2x2,3x3)
]


@pytest.mark.parametrize('case', WGRAD_CASES, ids=lambda c: 'B%d_H%d_%dto%d_k%d_form%d' % (c[0], c[1], c[2], c[3], c[4], c[6]))
def test_wgrad_with_batchnorm_relu_on_load_matches_fp64(L, case):
    B, H, Cin, Cout, k, pad, form = case
    y, sc, sh, _, a = _operands(B, H, Cin, Cout, k, 900 + Cin + Cout + B + form)
    dy = torch.randn(B, H, H, Cout, generator=torch.Generator().manual_seed(11 + B)).cuda()
    T = B * (H // 2) ** 2
    scratch = torch.empty(36 * Cout * Cin + 16 * T * (Cin + Cout), device='cuda') if form else None
    dw = torch.full((Cout, k, k, Cin), float('nan'), device='cuda')
    L.lib.call('simq_conv2d_wgrad_bnrelu_in', L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(dy), L.ptr(dw), B, H, H, Cin, Cout, k, k, 1, pad, form,
               L.ptr(scratch), L.stream_ptr())
    ref = torch.nn.grad.conv2d_weight(a.permute(0, 3, 1, 2), (Cout, Cin, k, k), dy.double().cpu().permute(0, 3, 1, 2), padding=pad).permute(0, 2, 3, 1)
    assert torch.isfinite(dw).all()
    err = rel(dw, ref)
    print('\nwgrad over relu(bn(y)) %d->%d k%d B=%d form %d: %.3g vs fp64' % (Cin, Cout, k, B, form, err))
    assert err < (1e-4 if (form == 1 and B % 4 == 0) else 1e-5)


def _train_once(options, precision, batch, cin=5, cout=2, seed=3):
    """One TD step (train.py:108-141) from seeded weights: reported loss / TD error, Q-map, TD targets, clipped gradient, BatchNorm buffers."""
    import simq
    import simq.learner as sl
    from oracle import cases, fcn as ofcn
    from simq import synth
    policy = simq.FCN(cin, cout, precision=precision, options=options)
    target = simq.FCN(cin, cout, precision=precision, options=options)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed)))
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed + 1)))
    policy.train(); target.eval()
    b = cases.make_batch(cin, cout, batch, 7)
    info = sl.train_step(policy, target, b, cases.GAMMA, batch, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP, use_double_dqn=True)
    torch.cuda.synchronize()
    return dict(info=info, q=policy._last['q'].clone(), y=policy._last['y'].clone(), g=policy.flat_grads.clone(), bn=policy.bn_buffers.clone(),
                options=dict(policy.plan.options))


@pytest.mark.parametrize('precision,batch', [('fp32', 6), ('fp32', 33), ('bf16', 8)], ids=['fp32_b6', 'fp32_b33', 'bf16_b8'])
def test_network_step_is_the_same_with_and_without_the_fusion(precision, batch):
    """fuse_bn1_apply (fp32) / bn1_mask_from_preact (bf16) change WHERE the BatchNorm-1 activation and its mask are formed, not the
    arithmetic: Q-map, TD targets, loss, BatchNorm buffers and the gradient of one TD step agree between the two settings to the
    round-off of the atomically accumulated sums (every kernel's arithmetic is otherwise the same on both sides)."""
    on = _train_once({'fuse_bn1_apply': 1, 'bn1_mask_from_preact': 1}, precision, batch)
    off = _train_once({'fuse_bn1_apply': 0, 'bn1_mask_from_preact': 0}, precision, batch)
    assert on['options']['fuse_bn1_apply'] == 1 and off['options']['fuse_bn1_apply'] == 0 and off['options']['bn1_mask_from_preact'] == 0
    fp32 = precision == 'fp32'
    # bf16: the forwards are the same kernels on both sides (the option only touches backward), fp32: the same values reach the same MFMAs
    assert rel(on['q'], off['q']) < (2e-6 if fp32 else 1e-6) and rel(on['y'], off['y']) < (2e-6 if fp32 else 1e-6)
    assert rel(on['bn'], off['bn']) < 1e-6
    assert abs(on['info']['loss'] - off['info']['loss']) <= 1e-5 * abs(off['info']['loss'])
    gerr = float((on['g'].double() - off['g'].double()).norm() / off['g'].double().norm())
    print('\n%s B=%d: loss %.8g | %.8g, gradient rel-L2 difference with / without the fusion %.3g' % (precision, batch, on['info']['loss'], off['info']['loss'], gerr))
    assert gerr < 5e-3       # (the gradient of these nets amplifies the 1e-7 summation-order noise of the statistics by ~1e4: tests/test_gpu_fcn.py header)
