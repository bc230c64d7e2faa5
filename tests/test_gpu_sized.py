"""GPU: BASELINE configs[2..4]'s per-GPU shapes against the REFERENCE at full size.

tests/golden/train_c5o2_b128.npz (configs[2] and configs[4]: Cin 5, Cout 2, 128 transitions per GPU), train_c5o2_b64.npz and
train_c5o1_b64.npz (configs[3]: lifting Cout 2 + pushing Cout 1, 64 per GPU and net) were written by oracle/gen_golden.py
`train_sized` from the imported reference: two consecutive train.train calls in fp32 (the oracle asserted bit-exact against them),
the fp64 oracle as the yardstick, the error of the reference's OWN fp32 on that batch, and the error of the reference's own modules
under torch.autocast('cpu', bfloat16) on that batch.  These are the batch sizes at which the large-batch kernels are selected (fp32:
the tile-menu entries and 36-plane Winograd problems of B >= 64; bf16: the image-tile implicit GEMM, the ping-pong weight gradient,
the 144x64 / 288-row LDS-DMA tiles) -- the network-level parity of those kernels against the reference is what this file adds.

Bars:
  fp32   loss / td error / q_sa / TD targets of step 1: 1e-4 against the reference's fp32 (and the fp64 yardstick; measured 2e-7 .. 1e-5);
         pre-clip gradient and first parameter update on the fixture's sampled elements against fp64: <= 3 x the error the
         reference's own fp32 makes on the same elements.  Measured at HEAD of round 3 / 4 (gpurun_out/t_all.log): train_c5o2_b128 2.6 x
         (5.7e-3 against the reference's 2.2e-3), train_c5o2_b64 1.5 x, train_c5o1_b64 1.1 x -- SINGLE batches of a fat-tailed distribution: the
         twelve-batch study at the same shape (tests/golden/grad_study_b128.npz, round 4, tests/test_gpu_fcn.py::
         test_gradient_parity_distribution[b128x12]) measures a median of 1.5 x, eleven cases within 2 x and one at 6.8 x whose cause is a
         single ReLU-mask flip under the one-hot gradient (tests/diag/diag_b128_maskflip.py), and is judged at median <= 2 x, at most one
         case beyond 3 x, none beyond 10 x;
         gradient norm 1e-3 (measured 1e-5 .. 5e-5); the second step per transition (q_sa, TD targets) at 1e-4 against the fp64 oracle run
         from the HIP path's own post-step-1 state (tests/step2_oracle.py: the loss against the fp64 trajectory is chaotic,
         2-7 % between fp32 summation orders on the b64 fixture); post-second-step parameter norms 1e-4; BatchNorm buffers 1e-4 after step 1
         (against fp64), 5e-3 after step 2 (they have seen the first update).
  bf16   against fp64, calibrated by the reference under bf16 autocast AT THIS SIZE (the fixture's bf16cal_* fields): loss, td error,
         q_sa <= 2 x calibration (measured 0.8 x, 0.9 x, 0.7-1.2 x).  The TD-loss gradient is NOT a usable yardstick for bf16 at any batch
         size: the reference's own autocast gradient is 0.44-0.49 off the fp64 gradient at B = 64-128 (double-DQN greedy actions flip --
         TD targets off by up to 0.7 --, ReLU masks move, the one-hot gradient cancels in every train-mode BatchNorm), the HIP bf16
         path measures 0.41 / 0.58 / 0.49 on the same batches; it is held to 1.5 x the calibration only as a guard against gross
         breakage.  The bf16 gradient bars that DO bind are (i) the dense-upstream-gradient case below, where the same backward
         walk is well conditioned (reference autocast ~1e-2), and (ii) per kernel on bf16-rounded operands against fp64 at 2e-5
         (tests/test_gpu_ops.py, including B = 64-128).

Dense upstream gradient (tests/golden/dense_*.npz, oracle/gen_golden.py `dense_grad`: the reference's own networks.FCN, loss =
sum(Q * R) for a seeded dense R, fp64 yardstick + the reference's fp32 and bf16-autocast errors).  Generated to find out whether the
one-hot form of the TD gradient is what makes these gradients ill-conditioned: it is not -- with a dense upstream gradient the
reference's own fp32 gradient is still 4.4-5.0e-3 off fp64 and its bf16-autocast gradient 0.41-0.60, at B = 64-128 (the train-mode
BatchNorm / ReLU chain of the randomly initialised network amplifies round-off by ~1e4 whatever the loss).  So:
  fp32   Q checksums 1e-4; sampled gradient rel-L2 and the worst per-tensor norm error <= 3 x the reference's own fp32 error;
  bf16   Q checksums <= 2 x, sampled gradient and per-tensor norms <= 1.5 x the reference's autocast error.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import cases
from oracle import fcn as ofcn
from oracle import learner as olearner
from simq import synth

import step2_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def simq_mod():
    import simq
    from simq import _lib  # noqa: F401
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return simq


def make_net(simq_mod, cin, cout, seed, training, precision, options=None):
    net = simq_mod.FCN(cin, cout, precision=precision, options=options)
    net.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed)))
    net.train(training)
    return net


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def rl2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def run_two_steps(simq_mod, case, precision, profile=False, options=None):
    """Two consecutive simq.train calls on the fixture's batch; returns scalars, the pre-clip gradient and first update on the
    fixture's sampled elements (reference OIHW indexing), per-tensor gradient norms, post-step summaries."""
    from simq._lib import lib
    name, cin, cout, B, wseed, dseed = case
    cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
    policy, target = make_net(simq_mod, cin, cout, wseed, True, precision, options), make_net(simq_mod, cin, cout, wseed + 1000, False, precision, options)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    p0 = [v.detach().clone().cpu().double() for v in policy.reference_views(policy.flat_params)]
    if profile:
        lib.call('simq_profile_start')
    info1 = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    kinds = None
    if profile:
        out = (ctypes.c_double * 12)()
        lib.call('simq_profile_stop', out, 3)
        kinds = [out[0], out[4], out[8]]
    tn = float(policy._simq_opt_state.total_norm.item())
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    grads = [v.detach().cpu().double() / coef for v in policy.reference_views(policy.flat_grads)]
    p1 = [v.detach().cpu().double() for v in policy.reference_views(policy.flat_params)]
    q_sa, y = policy._last['q_sa'].cpu().numpy(), policy._last['y'].cpu().numpy()
    bnvec = lambda sd_: np.concatenate([sd_[k].detach().double().cpu().numpy().ravel() for k in sd_
                                        if k.endswith('running_mean') or k.endswith('running_var')])
    bn1 = bnvec(policy.state_dict())
    sd1, sd_target = step2_oracle.snapshot(policy), step2_oracle.snapshot(target)
    info2 = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    q_sa2, y2 = policy._last['q_sa'].cpu().numpy(), policy._last['y'].cpu().numpy()
    gs, ds = [], []
    for t, a, b in zip(grads, p0, p1):
        idx = torch.tensor(cases.sample_indices(t.numel()))
        gs.append(t.reshape(-1)[idx].numpy())
        ds.append((b - a).reshape(-1)[idx].numpy())
    sd = policy.state_dict()
    p2 = [sd[k].detach().double().cpu() for k, _, kind in ofcn.state_spec(cin, cout) if ofcn.is_parameter(kind)]   # (incl. the unused fc)
    bn = bnvec(sd)
    return dict(info=[info1, info2], total_norm=tn, grad=np.stack(gs), dparam=np.stack(ds), gnorm=np.array([float(t.norm()) for t in grads]),
                q_sa=q_sa, y=y, p2_l2=np.array([float(t.norm()) for t in p2]), bn=bn, bn1=bn1, kinds=kinds,
                sd1=sd1, sd_target=sd_target, q_sa2=q_sa2, y2=y2, batch=batch,
                nbt=[int(sd[k]) for k in sd if k.endswith('num_batches_tracked')])


@pytest.mark.parametrize('case', cases.TRAIN_CASES_SIZED, ids=[c[0] for c in cases.TRAIN_CASES_SIZED])
def test_fp32_step_at_config_size_matches_the_reference(simq_mod, golden_dir, case):
    g = np.load('%s/%s.npz' % (golden_dir, case[0]))
    r = run_two_steps(simq_mod, case, 'fp32')
    e = dict(loss=abs(r['info'][0]['loss'] - float(g['loss'][0])) / float(g['loss'][0]),
             td=abs(r['info'][0]['td_error'] - float(g['td_error'][0])) / float(g['td_error'][0]),
             q_sa=relmax(r['q_sa'], g['q_sa']), y=relmax(r['y'], g['y']), q_sa64=relmax(r['q_sa'], g['q_sa64']),
             grad=rl2(r['grad'], g['grad64']), dparam=rl2(r['dparam'], g['dparam64']),
             norm=abs(r['total_norm'] - float(g['total_norm64'])) / float(g['total_norm64']),
             loss2=abs(r['info'][1]['loss'] - float(g['loss64'][1])) / float(g['loss64'][1]))
    big = g['grad_norm64'] > 1e-3 * float(g['total_norm64'])
    e['tensor_norms'] = float(np.abs(r['gnorm'][big] - g['grad_norm64'][big]).max() / g['grad_norm64'][big].max())
    print('\n[%s fp32] step-1 loss %.2g td %.2g q_sa %.2g (vs fp64 %.2g; reference fp32 %.2g) y %.2g | sampled gradient vs fp64 %.3g '
          '(reference fp32 %.3g), update %.3g (ref %.3g), |g| %.2g, per-tensor norms %.2g | step-2 loss vs fp64 %.3g (ref %.3g)'
          % (case[0], e['loss'], e['td'], e['q_sa'], e['q_sa64'], float(g['ref_q_err']), e['y'], e['grad'], float(g['ref_grad_err']), e['dparam'],
             float(g['ref_dparam_err']), e['norm'], e['tensor_norms'], e['loss2'], float(g['ref_loss_err'][1])))
    assert e['loss'] < 1e-4 and e['td'] < 1e-4 and e['q_sa'] < 1e-4 and e['y'] < 1e-4 and e['q_sa64'] < 1e-4
    # The sampled-gradient bar is asserted on a DETERMINISTIC plan (simq_plan_options.deterministic: fixed-order weight-gradient sums, so
    # the number is the same on every run and every box).  The default plan adds 16 of its 72 weight-gradient tensors with fp32 atomics:
    # its single-batch number (printed above) moves by +-15 % from run to run around a 2.6 x measurement of a 3 x bar at b128 -- it is
    # judged as a distribution by tests/test_gpu_fcn.py::test_gradient_parity_distribution[b128x12] (median <= 2 x), not here.
    rd = run_two_steps(simq_mod, case, 'fp32', options={'deterministic': 1})
    ed = dict(grad=rl2(rd['grad'], g['grad64']), dparam=rl2(rd['dparam'], g['dparam64']))
    print('[%s fp32] deterministic plan: sampled gradient vs fp64 %.3g (%.2f x the reference-fp32 error), update %.3g (%.2f x); default plan above: %.2f x / %.2f x'
          % (case[0], ed['grad'], ed['grad'] / float(g['ref_grad_err']), ed['dparam'], ed['dparam'] / float(g['ref_dparam_err']),
             e['grad'] / float(g['ref_grad_err']), e['dparam'] / float(g['ref_dparam_err'])))
    assert ed['grad'] <= 3.0 * float(g['ref_grad_err']), (ed['grad'], float(g['ref_grad_err']))
    assert ed['dparam'] <= 3.0 * float(g['ref_dparam_err']), (ed['dparam'], float(g['ref_dparam_err']))
    assert e['grad'] <= 10.0 * float(g['ref_grad_err']) and e['dparam'] <= 10.0 * float(g['ref_dparam_err'])      # (default plan: gross-breakage guard only)
    assert e['norm'] < 1e-3 and e['tensor_norms'] < 2e-2
    # the second step: per transition against the fp64 oracle started from the HIP path's own post-step-1 state, and the reported
    # loss against the Huber loss of those per-transition values (the loss against the fp64 TRAJECTORY is printed above only)
    e_q2, e_y2, ties = step2_oracle.second_step_against_the_oracle(r['sd1'], r['sd_target'], r['batch'], r['q_sa2'], r['y2'], r['info'][1])
    print('[%s fp32] step 2 against the oracle from the post-step-1 state: q_sa %.2g, TD targets %.2g (%d greedy-action ties)' % (case[0], e_q2, e_y2, ties))
    assert relmax(r['p2_l2'], g['param_summary_after2'][:, 1]) < 1e-4
    # running statistics: after the FIRST step (two train-mode forwards of the unmodified parameters) 1e-4 against fp64; after the
    # second they have seen the first update, whose fp32 error differs between implementations (the reference's included)
    assert relmax(r['bn1'], g['bn_buffers_after1_64']) < 1e-4
    assert relmax(r['bn'], g['bn_buffers_after2']) < 5e-3
    assert r['nbt'] == [int(v) for v in g['num_batches_tracked']]


@pytest.mark.parametrize('case', cases.TRAIN_CASES_SIZED, ids=[c[0] for c in cases.TRAIN_CASES_SIZED])
def test_bf16_step_at_config_size_within_the_reference_autocast_calibration(simq_mod, golden_dir, case):
    g = np.load('%s/%s.npz' % (golden_dir, case[0]))
    r = run_two_steps(simq_mod, case, 'bf16', profile=True)
    e = dict(loss=abs(r['info'][0]['loss'] - float(g['loss64'][0])) / float(g['loss64'][0]),
             td=abs(r['info'][0]['td_error'] - float(g['td_error64'][0])) / float(g['td_error64'][0]),
             q_sa=relmax(r['q_sa'], g['q_sa64']), y=relmax(r['y'], g['y64']),
             grad=rl2(r['grad'], g['grad64']), dparam=rl2(r['dparam'], g['dparam64']),
             norm=abs(r['total_norm'] - float(g['total_norm64'])) / float(g['total_norm64']))
    cal = {k: float(g['bf16cal_' + k]) for k in ('loss', 'td_error', 'q_sa', 'y', 'grad_sampled', 'dparam_sampled')}
    print('\n[%s bf16] vs fp64: loss %.3g (reference autocast %.3g) td %.3g (%.3g) q_sa %.3g (%.3g) y %.3g (%.3g) | sampled gradient %.3g (%.3g) '
          'update %.3g (%.3g) |g| %.3g | launches: image-tile/dominant %d, direct wgrad %d, other tiles %d'
          % (case[0], e['loss'], cal['loss'], e['td'], cal['td_error'], e['q_sa'], cal['q_sa'], e['y'], cal['y'], e['grad'], cal['grad_sampled'],
             e['dparam'], cal['dparam_sampled'], e['norm'], r['kinds'][0], r['kinds'][1], r['kinds'][2]))
    assert all(np.isfinite(i['loss']) for i in r['info'])
    assert e['loss'] <= 2.0 * cal['loss'] and e['td'] <= 2.0 * cal['td_error'] and e['q_sa'] <= 2.0 * cal['q_sa']
    assert e['grad'] <= 1.5 * cal['grad_sampled'] and e['dparam'] <= 1.5 * cal['dparam_sampled']
    assert r['kinds'][1] > 0 and r['kinds'][0] + r['kinds'][2] > 0
    if case[3] >= 128:
        assert r['kinds'][0] > 0, 'the image-tile kernel (the bench leg\'s dominant kernel) was not selected at B = %d' % case[3]


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', cases.DENSE_GRAD_CASES, ids=[c[0] for c in cases.DENSE_GRAD_CASES])
def test_dense_gradient_backward_at_config_size(simq_mod, golden_dir, case, precision):
    """Forward (train-mode BatchNorm) + backward of a DENSE upstream gradient at 128 / 64 samples against the reference's own FCN
    (fixture from oracle/gen_golden.py dense_grad): the well-conditioned network-level gradient check -- every large-batch forward,
    dgrad and weight-gradient kernel contributes, and a kernel that is a few per cent off moves the result by orders of magnitude
    more than the bar."""
    from simq._lib import MODE_TRAIN
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    net = make_net(simq_mod, cin, cout, wseed, True, precision)
    x = torch.from_numpy(synth.make_states(B, cin, dseed)).cuda()
    R = torch.from_numpy(cases.dense_upstream(cout, B, dseed)).cuda()
    q = net._forward_raw(x, MODE_TRAIN)
    net._backward_raw(R.contiguous(), B)
    grads = [v.detach().cpu().double() for v in net.reference_views(net.flat_grads)]
    qd = q.double()
    qsum = np.array([float(qd.sum()), float(qd.abs().sum()), float((qd * R.double()).sum())])
    samp = np.stack([t.reshape(-1)[torch.tensor(cases.sample_indices(t.numel()))].numpy() for t in grads])
    norms = np.array([float(t.norm()) for t in grads])
    err = rl2(samp, g['grad64'])
    big = g['grad_norm64'] > 1e-3 * g['grad_norm64'].max()
    worst = float((np.abs(norms - g['grad_norm64'])[big] / g['grad_norm64'][big]).max())
    qerr = float(np.abs(qsum[1:] - g['q_checksum64'][1:]).max() / np.abs(g['q_checksum64'][1:]).max())
    ref = {k: float(g[k]) for k in ('ref_fp32_grad_sampled', 'ref_fp32_worst_tensor', 'bf16cal_grad_sampled', 'bf16cal_worst_tensor', 'bf16cal_q', 'ref_fp32_q')}
    print('\n[%s %s] vs fp64: Q checksums %.3g | sampled gradient rel-L2 %.3g (reference fp32 %.3g, reference bf16-autocast %.3g) | worst '
          'per-tensor norm error %.3g (reference fp32 worst-tensor rel-L2 %.3g, autocast %.3g)'
          % (name, precision, qerr, err, ref['ref_fp32_grad_sampled'], ref['bf16cal_grad_sampled'], worst, ref['ref_fp32_worst_tensor'],
             ref['bf16cal_worst_tensor']))
    if precision == 'fp32':
        assert qerr < 1e-4
        assert err <= 3.0 * ref['ref_fp32_grad_sampled'], (err, ref['ref_fp32_grad_sampled'])
        assert worst <= 3.0 * ref['ref_fp32_worst_tensor'], (worst, ref['ref_fp32_worst_tensor'])
    else:
        assert qerr <= max(2.0 * ref['bf16cal_q'], 2e-2)
        assert err <= 1.5 * ref['bf16cal_grad_sampled'], (err, ref['bf16cal_grad_sampled'])
        assert worst <= 1.5 * ref['bf16cal_worst_tensor'], (worst, ref['bf16cal_worst_tensor'])
