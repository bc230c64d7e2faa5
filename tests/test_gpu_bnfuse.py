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


# ---- the elementwise BatchNorm passes on their own (simq_bn_relu_apply / simq_bn_relu_backward), fp32 and all-bf16 forms --------------

def _bf16(t):
    return t.to(torch.bfloat16)


def _bn_case(rows, C, storage, seed, residual):
    g = torch.Generator().manual_seed(seed)
    acc = torch.randn(rows, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + torch.randn(C, generator=g)     # the convolution's fp32 accumulators
    stats = torch.cat([acc.double().sum(0), (acc.double() ** 2).sum(0)])                                        # what its epilogue leaves
    y = _bf16(acc) if storage else acc                                                                          # what it stores
    gamma, beta = 0.5 + torch.rand(C, generator=g), 0.4 * torch.randn(C, generator=g)
    res = torch.randn(rows, C, generator=g) if residual else None
    if res is not None and storage:
        res = _bf16(res)
    mean = stats[:C] / rows
    var = (stats[C:] / rows - mean * mean).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    return acc, stats, y, gamma, beta, res, mean, var, invstd


@pytest.mark.parametrize('storage', [0, 1], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('rows,C,residual', [(3 * 576, 64, False), (5 * 576, 256, True), (2 * 576, 512, True), (7 * 576, 128, False)],
                         ids=['c64', 'c256_res', 'c512_res', 'c128'])
def test_bn_relu_apply_against_fp64(L, storage, rows, C, residual):
    """bn_apply_kernel / bn_apply16_kernel: relu(bn(y) [+ res]) with statistics from the unrounded accumulators, applied to the stored
    value.  fp32: 2e-6 of the range against fp64.  bf16: the output plane is the fp64 result rounded to bf16 except where fp32-vs-fp64
    arithmetic crosses a rounding boundary (one ulp, a handful of elements); saved scale / shift / mean / invstd and the running update
    (momentum 0.1, unbiased variance) at 1e-6."""
    acc, stats, y, gamma, beta, res, mean, var, invstd = _bn_case(rows, C, storage, 40 + C + storage, residual)
    ref = (y.double() - mean) * invstd * gamma.double() + beta.double()
    if res is not None:
        ref = ref + res.double()
    ref = torch.relu(ref)
    running = torch.cat([torch.full((C,), 0.25), torch.full((C,), 1.5)]).cuda()
    saved = torch.zeros(4 * C, device='cuda')
    if storage:
        out = torch.zeros(rows, C, dtype=torch.bfloat16, device='cuda')
    else:
        out = torch.full((rows, C), float('nan'), device='cuda')
    yd, rd, sd, gd, bd = y.cuda(), (res.cuda() if res is not None else None), stats.cuda(), gamma.cuda(), beta.cuda()    # (kept alive across the launch)
    L.lib.call('simq_bn_relu_apply', L.ptr(yd), L.ptr(sd), L.ptr(gd), L.ptr(bd), L.ptr(rd), 1, L.ptr(out), rows, C, storage,
               L.ptr(saved), L.ptr(running), L.stream_ptr())
    o = out.float().cpu().double()
    if storage:
        want = ref.to(torch.bfloat16).double()
        off = (o != want)
        frac = float(off.double().mean())
        ulp = float(((o - want).abs() / want.abs().clamp_min(1e-3)).max())
        print('\nbn_apply16 C=%d: %.4f %% of the elements one rounding away from bf16(fp64 result), largest %.3g relative' % (C, 100 * frac, ulp))
        assert frac < 2e-3 and ulp < 2 ** -7
    else:
        assert rel(o, ref) < 2e-6
    s = saved.cpu().double()
    scale = gamma.double() * invstd
    assert rel(s[:C], scale) < 1e-6 and rel(s[C:2 * C], beta.double() - mean * scale) < 1e-6 and rel(s[2 * C:3 * C], mean) < 1e-6 and rel(s[3 * C:], invstd) < 1e-6
    r = running.cpu().double()
    assert rel(r[:C], 0.9 * 0.25 + 0.1 * mean) < 1e-6 and rel(r[C:], 0.9 * 1.5 + 0.1 * var * rows / (rows - 1)) < 1e-6


@pytest.mark.parametrize('storage', [0, 1], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('mask_kind', [0, 1, 2], ids=['nomask', 'mask_tensor', 'mask_from_preact'])
@pytest.mark.parametrize('rows,C', [(3 * 576, 64), (4 * 576, 256), (2 * 576, 512)], ids=['c64', 'c256', 'c512'])
def test_bn_relu_backward_against_fp64(L, storage, mask_kind, rows, C):
    """bn_bwd_apply_kernel / bn_bwd_apply16_kernel<MFY>: dz = g * mask, dy = gamma * invstd * (dz - sum(dz)/rows - xhat * sum(dz * xhat)/rows)
    with the two sums given (the dgrad epilogue's job in the plan).  The mask as a tensor (fp32 activation / bf16 plane) and RECOMPUTED from
    the pre-BN output (scale * y + shift > 0: simq_plan_options.fuse_bn1_apply / bn1_mask_from_preact) must select the same elements."""
    acc, stats, y, gamma, beta, _, mean, var, invstd = _bn_case(rows, C, storage, 60 + C + storage + mask_kind, False)
    gen = torch.Generator().manual_seed(5 + C)
    g = torch.randn(rows, C, generator=gen)
    if storage:
        g = _bf16(g)
    saved = torch.zeros(4 * C, device='cuda')
    running = torch.zeros(2 * C, device='cuda')
    act = torch.zeros(rows, C, dtype=torch.bfloat16 if storage else torch.float32, device='cuda')
    yd, sd, gd, bd, gg = y.cuda(), stats.cuda(), gamma.cuda(), beta.cuda(), g.cuda()                                     # (kept alive across the launches)
    L.lib.call('simq_bn_relu_apply', L.ptr(yd), L.ptr(sd), L.ptr(gd), L.ptr(bd), None, 1, L.ptr(act), rows, C, storage,
               L.ptr(saved), L.ptr(running), L.stream_ptr())
    s = saved.cpu()
    pre = torch.addcmul(s[C:2 * C], y.float(), s[:C])                       # fma(y, scale, shift) as the kernels form it (fp32)
    mask = (act.float().cpu() > 0) if mask_kind else torch.ones(rows, C, dtype=torch.bool)
    if mask_kind:
        agree = float(((pre > 0) == mask).double().mean())
        assert agree > 1 - 1e-5, 'recomputed mask differs from the stored activation on %.3g of the elements' % (1 - agree)
    dz = torch.where(mask, g.double(), torch.zeros((), dtype=torch.float64))
    xhat = (y.double() - s[2 * C:3 * C].double()) * s[3 * C:].double()
    red = torch.cat([dz.sum(0), (dz * xhat).sum(0)])
    ref = gamma.double() * s[3 * C:].double() * (dz - red[:C] / rows - xhat * red[C:] / rows)
    dt = torch.bfloat16 if storage else torch.float32
    dy = torch.zeros(rows, C, dtype=dt, device='cuda')
    dzo = torch.zeros(rows, C, dtype=dt, device='cuda')
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    redd = red.cuda()
    L.lib.call('simq_bn_relu_backward', L.ptr(gg), L.ptr(act) if mask_kind == 1 else None, mask_kind, L.ptr(yd), L.ptr(saved), L.ptr(gd),
               L.ptr(redd), L.ptr(dy), L.ptr(dzo), L.ptr(dgam), L.ptr(dbet), rows, C, storage, L.stream_ptr())
    o = dy.float().cpu().double()
    assert torch.equal(dzo.float().cpu().double(), dz), 'the masked gradient is exact in either storage'
    assert rel(dgam, red[C:]) < 1e-6 and rel(dbet, red[:C]) < 1e-6
    if storage:
        want = ref.to(torch.bfloat16).double()
        frac = float((o != want).double().mean())
        worst = float(((o - want).abs() / want.abs().clamp_min(1e-2)).max())
        print('\nbn_bwd_apply16 C=%d mask %d: %.4f %% of the elements one rounding away from bf16(fp64 result), largest %.3g' % (C, mask_kind, 100 * frac, worst))
        assert frac < 5e-3 and worst < 2 ** -7
    else:
        assert rel(o, ref) < 5e-6


@pytest.mark.parametrize('precision,batch', [('fp32', 32), ('bf16', 64), ('fp32', 5)], ids=['fp32_b32', 'bf16_b64', 'fp32_b5'])
def test_deterministic_plans_repeat_bit_for_bit(precision, batch):
    """simq_plan_options.deterministic = 1: two TD steps from identical state give bit-identical Q-maps, TD targets, loss sums, BatchNorm
    buffers, gradients and updated parameters -- the pixel-split weight-gradient kernels leave per-split slabs summed in split order instead of
    fp32 atomics, the one-hot head backward walks the transitions in order.  (Default plans repeat everything but those weight gradients:
    tests/diag/diag_determinism.py.)  The deterministic gradient is the default plan's gradient to summation-order round-off."""
    a = _train_once({'deterministic': 1}, precision, batch)
    b = _train_once({'deterministic': 1}, precision, batch)
    assert a['options']['deterministic'] == 1
    assert a['info'] == b['info']
    for k in ('q', 'y', 'bn', 'g'):
        assert torch.equal(a[k], b[k]), 'deterministic plan: %s differs between two identical runs' % k
    c = _train_once({}, precision, batch)
    assert torch.equal(a['q'], c['q']) and torch.equal(a['y'], c['y'])
    gerr = float((a['g'].double() - c['g'].double()).norm() / c['g'].double().norm())
    print('\n%s B=%d: deterministic vs default gradient rel-L2 %.3g' % (precision, batch, gerr))
    assert gerr < 1e-5
