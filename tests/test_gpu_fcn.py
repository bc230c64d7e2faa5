"""GPU: the HIP Q-network / learner against the CPU oracle and the committed golden fixtures.

Parity bars (BASELINE.json north_star + SURVEY section 0):
  * Q-maps, loss, td_error, q_sa, TD targets, BN statistics: <= 1e-4 (max-abs err / max-abs value)
    vs the fp32 oracle AND vs the golden vectors written from the imported reference;
  * gradients / post-step weights: train-mode BN backward of a one-hot upstream gradient cancels
    catastrophically and every ReLU whose pre-activation is within round-off of 0 flips its mask
    (tests/diag/diag_grad_error.py: one flipped element in the head is a 1e-4 error everywhere below it),
    so the reference's OWN fp32 gradient is only 3e-4..2e-2 accurate vs fp64
    (tests/golden/train_*.npz: ref_fp32_grad_relerr; which implementation is luckier varies per
    batch: on the three fixture batches the reference-fp32 / HIP-fp32 errors are 3.4e-4 / 8.0e-3,
    1.8e-2 / 4.6e-4 and 2.1e-3 / 1.5e-4 -- tests/diag/diag_grad_error.py).  The HIP gradient is therefore
    judged against the fp64 oracle with ONE bar for the whole fp32 class: global relative L2 error
    <= 5e-2 (2.5 x the worst error of the reference's own fp32); the per-kernel backward ops are held
    to 1e-4 in test_gpu_ops.py, where no such conditioning is involved.
"""
import copy
import random
import types

import numpy as np
import pytest
import torch
from torch.nn.functional import smooth_l1_loss

from oracle import cases
from oracle import fcn as ofcn
from oracle import learner as olearner
from simq import arch, synth

import step2_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope='module')
def simq_mod():
    import simq
    from simq import _lib  # noqa: F401  (raises loudly when libsimq.so is missing)
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return simq


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make_net(simq_mod, cin, cout, seed, training, precision='fp32', options=None):
    net = simq_mod.FCN(cin, cout, precision=precision, options=options)
    net.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed)))
    net.train(training)
    return net


def grads_to_reference_layout(net):
    """flat HIP gradient buffer -> {reference key: tensor in reference (OIHW) layout} on the CPU."""
    out = {}
    g = net.flat_grads.detach().cpu()
    for (name, _, kind), (off, n, shape) in zip(net._param_names, net._grad_views):
        t = g[off:off + n].view(shape)
        if len(shape) == 4:
            t = t.permute(0, 3, 1, 2).contiguous()
        out[arch.PREFIX + name] = t
    return out


def global_rel_l2(got, ref):
    num = sum(float((got[k].double() - ref[k].double()).pow(2).sum()) for k in ref)
    den = sum(float(ref[k].double().pow(2).sum()) for k in ref)
    return (num / den) ** 0.5


@pytest.mark.parametrize('case', cases.FORWARD_CASES, ids=[c[0] for c in cases.FORWARD_CASES])
def test_forward_eval_and_train(simq_mod, case, golden_dir):
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    x_hwc = synth.make_states(B, cin, dseed)
    x_nchw = torch.cat([olearner.apply_transform(s) for s in x_hwc])
    # eval mode: golden (reference) and live oracle, through the reference-layout entry point (NCHW)
    net = make_net(simq_mod, cin, cout, wseed, training=False)
    with torch.no_grad():
        q = net(x_nchw.cuda())
    assert tuple(q.shape) == (B, cout, 96, 96)
    assert rel(q, g['q_eval']) < TOL
    st = cases.oracle_state(cin, cout, wseed)
    with torch.no_grad():
        q_or = ofcn.fcn_forward(st, x_nchw, False)
    assert rel(q, q_or) < TOL
    assert all(v == 0 for v in net.num_batches_tracked.values())
    # argmax of every sample agrees with the oracle's (policies.py:64)
    assert q.reshape(B, -1).argmax(1).cpu().tolist() == q_or.reshape(B, -1).argmax(1).tolist()
    # train mode: batch statistics + running-stat update, HWC entry point (replay layout)
    net = make_net(simq_mod, cin, cout, wseed, training=True)
    with torch.no_grad():
        q = net.forward_nhwc(torch.from_numpy(x_hwc).cuda())
    assert rel(q, g['q_train']) < TOL
    sd = net.state_dict()
    got = np.concatenate([sd[k].cpu().double().numpy().ravel() for k in sd
                          if k.endswith('running_mean') or k.endswith('running_var')])
    assert rel(got, g['bn_buffers_after']) < TOL
    assert all(int(sd[k]) == 1 for k in sd if k.endswith('num_batches_tracked'))


@pytest.mark.parametrize('cin,cout,wseed,dseed', [(7, 2, 91, 92), (10, 1, 93, 94), (4, 2, 95, 96)], ids=['c7o2_spatial', 'c10o1', 'c4o2'])
def test_forward_b8_other_channel_counts_and_f4_forwards(simq_mod, cin, cout, wseed, dseed):
    """(1) The input-channel variants SURVEY 8d names beyond the fixtures: Cin = 7 (`-spatial`: 4 state + 3 spatial intention channels)
    and Cin = 10 (the largest num_input_channels the reference's configs produce), Cout 2 / 1.  (2) Batch 8 is where the no-grad
    forwards of the fp32 plan switch their wide 3x3 layers to Winograd F(4x4,3x3) (B * 36 >= 256 tiles): eval mode (target net,
    policy.step) and train-mode-no-grad (the double-DQN argmax forward, with batch statistics and the running update) against the
    live fp32 oracle at the 1e-4 bar, argmax included; the grad-mode forward (F(2x2,3x3)) of the same batch agrees as well."""
    B = 8
    x_hwc = synth.make_states(B, cin, dseed)
    x_nchw = torch.cat([olearner.apply_transform(s) for s in x_hwc])
    net = make_net(simq_mod, cin, cout, wseed, training=False)
    with torch.no_grad():
        q = net.forward_nhwc(torch.from_numpy(x_hwc).cuda())
    st = cases.oracle_state(cin, cout, wseed)
    with torch.no_grad():
        q_or = ofcn.fcn_forward(st, x_nchw, False)
    err_eval = rel(q, q_or)
    assert q.reshape(B, -1).argmax(1).cpu().tolist() == q_or.reshape(B, -1).argmax(1).tolist()
    net = make_net(simq_mod, cin, cout, wseed, training=True)
    with torch.no_grad():
        q_ng = net.forward_nhwc(torch.from_numpy(x_hwc).cuda())                   # MODE_TRAIN_NOGRAD
    st = cases.oracle_state(cin, cout, wseed)
    with torch.no_grad():
        q_or_t = ofcn.fcn_forward(st, x_nchw, True)
    err_ng = rel(q_ng, q_or_t)
    sd = net.state_dict()
    got = np.concatenate([sd[k].cpu().double().numpy().ravel() for k in sd if k.endswith('running_mean') or k.endswith('running_var')])
    err_bn = rel(got, cases.bn_buffer_vector(st))
    net2 = make_net(simq_mod, cin, cout, wseed, training=True)
    q_g = net2.forward_nhwc(torch.from_numpy(x_hwc).cuda())                       # grad mode (autograd node): F(2x2,3x3)
    err_g = rel(q_g, q_or_t)
    print('\n[Cin %d Cout %d B 8] Q-map error vs the fp32 oracle: eval (F4) %.3g, train-no-grad (F4) %.3g, grad-mode (F2) %.3g, running stats %.3g'
          % (cin, cout, err_eval, err_ng, err_g, err_bn))
    assert err_eval < TOL and err_ng < TOL and err_g < TOL and err_bn < TOL


def test_state_dict_roundtrip(simq_mod):
    np_sd = synth.make_state_dict(5, 2, 77)
    net = make_net(simq_mod, 5, 2, 77, training=False)
    sd = net.state_dict()
    assert list(sd.keys()) == list(np_sd.keys())
    for k, v in np_sd.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        assert np.array_equal(sd[k].cpu().numpy(), v), k
    assert len(list(net.parameters())) == 72
    assert sum(p.numel() for p in net.parameters()) == sum(
        int(np.prod(s)) for _, s, kd in arch.state_spec(5, 2) if kd in arch.TRAINABLE_KINDS + ('fc_w', 'fc_b'))
    other = simq_mod.FCN(5, 2)
    other.load_state_dict({k[len('module.'):]: v for k, v in sd.items()})    # un-prefixed keys load too
    assert torch.equal(other.flat_params, net.flat_params) and torch.equal(other.bn_buffers, net.bn_buffers)


@pytest.mark.parametrize('case', cases.TRAIN_CASES[:2] + cases.TRAIN_CASES_CIN, ids=[c[0] for c in cases.TRAIN_CASES[:2] + cases.TRAIN_CASES_CIN])
def test_autograd_path_reference_style_train(simq_mod, case, golden_dir):
    """The reference's own train() recipe (torch gather / smooth_l1 / backward / clip / optim.SGD,
    train.py:108-141) driving simq.FCN through autograd -- only the network is HIP."""
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    batch = cases.make_batch(cin, cout, B, dseed)
    policy = make_net(simq_mod, cin, cout, wseed, training=True)
    target = make_net(simq_mod, cin, cout, wseed + 1000, training=False)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    tf = olearner.apply_transform
    dev = torch.device('cuda')
    losses, tds, norms = [], [], []
    for _ in range(2):
        state_batch = torch.cat([tf(s) for s in batch.state]).to(dev)
        action_batch = torch.tensor(batch.action, dtype=torch.long).to(dev)
        reward_batch = torch.tensor(batch.reward, dtype=torch.float32).to(dev)
        nfns = torch.cat([tf(s) for s in batch.next_state if s is not None]).to(dev)
        output = policy(state_batch)
        q_sa = output.view(B, -1).gather(1, action_batch.unsqueeze(1)).squeeze(1)
        nsv = torch.zeros(B, dtype=torch.float32, device=dev)
        mask = torch.tensor(tuple(s is not None for s in batch.next_state), dtype=torch.bool, device=dev)
        with torch.no_grad():
            best = policy(nfns).view(nfns.size(0), -1).max(1)[1].view(-1, 1)
            nsv[mask] = target(nfns).view(nfns.size(0), -1).gather(1, best).view(-1)
        y = reward_batch + cases.GAMMA * nsv
        td = torch.abs(q_sa - y).detach()
        loss = smooth_l1_loss(q_sa, y)
        opt.zero_grad()
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(policy.parameters(), cases.CLIP)))
        opt.step()
        losses.append(loss.item())
        tds.append(td.mean().item())
    assert rel(losses[0], g['loss'][0]) < TOL and rel(tds[0], g['td_error'][0]) < TOL
    assert policy.fc_weight.grad is None and policy.fc_bias.grad is None
    sd = policy.state_dict()
    assert all(int(sd[k]) == 4 for k in sd if k.endswith('num_batches_tracked'))
    # total gradient norm: reference fp32 value is itself only ~ref_err accurate
    ref_err = float(g['ref_fp32_grad_relerr'])
    assert abs(norms[0] - float(g['total_norm64'])) <= 5e-2 * float(g['total_norm64'])


@pytest.mark.parametrize('cin,cout,B', [(4, 3, 3), (5, 4, 4)], ids=['c4o3_b3', 'c5o4_b4'])
def test_fused_train_with_three_and_four_output_channels_against_the_live_oracle(simq_mod, cin, cout, B):
    """networks.FCN takes any num_output_channels (networks.py:8-14); the reference's robot types use 1 or 2, libsimq's plans go to 4 (head
    kernels, Q-map reductions and the one-hot backward are written for Cout <= 4).  No golden fixture exists for 3 / 4: one train() call against
    the oracle run live in fp64 (loss and td error at 1e-4, the gradient at the bar of the golden cases)."""
    cfg = cases.make_cfg(B)
    batch = cases.make_batch(cin, cout, B, 1)
    spec = ofcn.state_spec(cin, cout)
    policy = make_net(simq_mod, cin, cout, 2, training=True)
    target = make_net(simq_mod, cin, cout, 1002, training=False)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    st, tg = cases.oracle_state(cin, cout, 2, torch.float64), cases.oracle_state(cin, cout, 1002, torch.float64)
    ex64 = {}
    ref = olearner.train_step(cfg, st, tg, spec, [None] * len(olearner.grad_keys(spec)), batch, cases.GAMMA, cases.LR, cases.MOMENTUM,
                              cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
    info = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    assert rel(info['loss'], ref['loss']) < TOL and rel(info['td_error'], ref['td_error']) < TOL
    assert policy._last['q'].shape[1] == cout
    tn = float(policy._simq_opt_state.total_norm.item())
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    got = {k: v / coef for k, v in grads_to_reference_layout(policy).items()}
    err = global_rel_l2(got, ex64['grads'])
    assert err <= 5e-2, 'gradient rel-L2 error %.3g vs fp64' % err


@pytest.mark.parametrize('case', cases.TRAIN_CASES + cases.TRAIN_CASES_CIN, ids=[c[0] for c in cases.TRAIN_CASES + cases.TRAIN_CASES_CIN])
def test_fused_train_vs_golden_and_oracle(simq_mod, case, golden_dir):
    """simq.train (drop-in signature of train.py:108) -- two consecutive calls.  train_c{3,6,7,10}*: the reference's other input-channel
    counts (tools_generate_experiments.py:200-204) -- the stem's forward and weight gradient address K = 49 * Cin; Cin = 10 leaves the
    dedicated fp32 stem kernel (7 * Cin <= 64) for the generic implicit GEMM."""
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    cfg = cases.make_cfg(B)
    batch = cases.make_batch(cin, cout, B, dseed)
    spec = ofcn.state_spec(cin, cout)
    policy = make_net(simq_mod, cin, cout, wseed, training=True)
    target = make_net(simq_mod, cin, cout, wseed + 1000, training=False)
    target_before = target.flat_params.clone()
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    # fp64 oracle for step 1 (gradient yardstick), fp32 oracle for two full steps
    st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
    ex64 = {}
    olearner.train_step(cfg, st64, tg64, spec, [None] * len(olearner.grad_keys(spec)), batch, cases.GAMMA, cases.LR,
                        cases.MOMENTUM, cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
    info1 = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    assert set(info1) == {'td_error', 'loss'} and all(isinstance(v, float) for v in info1.values())
    assert rel(info1['loss'], g['loss'][0]) < TOL and rel(info1['td_error'], g['td_error'][0]) < TOL
    assert rel(policy._last['q_sa'], g['q_sa']) < TOL and rel(policy._last['y'], g['y']) < TOL
    if g['output_step1'].size:
        assert rel(policy._last['q'], g['output_step1']) < TOL
    # gradients (clipped in place by clip coefficient c = min(1, 100/norm)) vs fp64
    ref_err = float(g['ref_fp32_grad_relerr'])
    bar = 5e-2
    tn = float(policy._simq_opt_state.total_norm.item())
    assert abs(tn - float(g['total_norm64'])) <= bar * float(g['total_norm64'])
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    got = {k: v / coef for k, v in grads_to_reference_layout(policy).items()}
    err = global_rel_l2(got, ex64['grads'])
    assert err <= bar, 'gradient rel-L2 error %.3g vs fp64 (reference fp32 itself: %.3g)' % (err, ref_err)
    # exactly-zero-in-theory gradients (a train-mode BN follows the head biases): absolute bar
    assert got['module.conv1.bias'].abs().max() < 1e-5 and got['module.conv2.bias'].abs().max() < 1e-5
    # momentum buffers live where torch.optim.SGD keeps them
    p0 = next(iter(policy.parameters()))
    assert opt.state[p0]['momentum_buffer'].data_ptr() == policy._simq_opt_state.momentum.data_ptr()
    sd1, sd_target = step2_oracle.snapshot(policy), step2_oracle.snapshot(target)
    info2 = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    # The second call: per transition at 1e-4 against the fp64 oracle run from the HIP path's own post-step-1 state (tests/
    # step2_oracle.py), which is what "the second train() call is right" means.  Against the golden TRAJECTORY only a sanity bound:
    # with these synthetic +-40 TD errors the clipped step has norm lr*100 = 1.0 and |grad| = 712, so a 1e-3 gradient round-off
    # difference moves the next loss by up to ~2 % (observed 0.1-5 % between fp32 implementations)
    step2_oracle.second_step_against_the_oracle(sd1, sd_target, batch, policy._last['q_sa'].cpu().numpy(), policy._last['y'].cpu().numpy(), info2)
    assert rel(info2['loss'], g['loss'][1]) < 0.1 and rel(info2['td_error'], g['td_error'][1]) < 0.1
    sd = policy.state_dict()
    assert all(int(sd[k]) == 4 for k in sd if k.endswith('num_batches_tracked'))     # 2 per train() call
    got_bn = np.concatenate([sd[k].cpu().double().numpy().ravel() for k in sd
                             if k.endswith('running_mean') or k.endswith('running_var')])
    # running statistics after the 2nd call see weights that already moved by lr * (ill-conditioned gradient)
    assert rel(got_bn, g['bn_buffers_after2']) < max(10 * TOL, bar)
    # post-step parameters: per-tensor (sum, L2) summary of the golden
    rows = []
    for k, _, kind in spec:
        if ofcn.is_parameter(kind):
            t = sd[k].double().cpu()
            rows.append([float(t.sum()), float(t.norm())])
    rows = np.asarray(rows)
    assert np.abs(rows[:, 1] - g['param_summary_after2'][:, 1]).max() <= 1e-4 * g['param_summary_after2'][:, 1].max()
    # the target net is never modified by train()
    assert torch.equal(target.flat_params, target_before)
    assert all(v == 0 for v in target.num_batches_tracked.values())


def test_vanilla_dqn_and_no_clip(simq_mod):
    cin, cout, B = 4, 2, 4
    cfg = cases.make_cfg(B, use_double_dqn=False, grad_norm_clipping=None)
    batch = cases.make_batch(cin, cout, B, 91)
    spec = ofcn.state_spec(cin, cout)
    policy, target = make_net(simq_mod, cin, cout, 92, True), make_net(simq_mod, cin, cout, 93, False)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    st, tg = cases.oracle_state(cin, cout, 92), cases.oracle_state(cin, cout, 93)
    ref = olearner.train_step(cfg, st, tg, spec, [None] * len(olearner.grad_keys(spec)), batch, cases.GAMMA, cases.LR,
                              cases.MOMENTUM, cases.WEIGHT_DECAY)
    got = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    assert rel(got['loss'], ref['loss']) < TOL and rel(got['td_error'], ref['td_error']) < TOL
    sd = policy.state_dict()
    assert all(int(sd[k]) == 1 for k in sd if k.endswith('num_batches_tracked'))     # vanilla DQN: one train fwd


def test_train_returns_the_loss_without_a_stream_synchronisation(simq_mod):
    """train() hands back loss / td_error of train.py:137-139 through the library's early copy (simq_train_args.loss_host +
    simq_train_loss_wait: issued right behind the TD / Huber launch, not behind backward + SGD).  The floats are the device-side sums
    of the same step, two nets trained alternately keep their own values, the sampler's packed asynchronous upload delivers the same
    minibatch as plain copies, and a wait without a pending copy is an error."""
    from simq import learner
    from simq._lib import lib
    cin, B = 5, 6
    nets = []
    for cout, seed in ((2, 31), (1, 33)):
        policy, target = make_net(simq_mod, cin, cout, seed, True), make_net(simq_mod, cin, cout, seed + 1, False)
        ring = simq_mod.DeviceReplayBuffer(64, cin)
        for t in synth.make_transitions(40, cin, cout, seed + 2, terminal_frac=0.2):
            ring.push(*t)
        nets.append((policy, target, ring))
    for step in range(3):
        for policy, target, ring in nets:
            idx = ring.sample_indices(B)
            batch = ring.gather(idx)
            # the packed upload against the records themselves
            recs = [ring.buffer[i] for i in idx]
            assert batch.action.tolist() == [r.action for r in recs]
            assert np.allclose(batch.reward.cpu().numpy(), np.asarray([r.reward for r in recs], np.float32), rtol=0, atol=0)
            assert batch.nonfinal_pos.tolist() == [i for i, r in enumerate(recs) if r.next_state is not None]
            assert batch.action.dtype == torch.int64 and batch.reward.dtype == torch.float32 and batch.nonfinal_pos.dtype == torch.int32
            info = learner.train_step(policy, target, batch, cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP)
            q_sa, y = policy._last['q_sa'].double().cpu().numpy(), policy._last['y'].double().cpu().numpy()
            d = q_sa - y
            huber = float(np.where(np.abs(d) < 1.0, 0.5 * d * d, np.abs(d) - 0.5).mean())
            assert abs(info['loss'] - huber) <= 1e-5 * huber and abs(info['td_error'] - np.abs(d).mean()) <= 1e-5 * np.abs(d).mean()
    torch.cuda.synchronize()
    assert lib.c.simq_train_loss_wait(nets[0][0].plan.handle) != 0 and 'no simq_train_step' in simq_mod._lib.last_error()


def test_policy_step_golden(simq_mod, golden_dir):
    g = np.load('%s/policy_step.npz' % golden_dir)
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}], num_input_channels=4,
                                final_exploration=0.01, checkpoint_path=None)
    pol = simq_mod.DQNPolicy(cfg, train=False, random_seed=5)
    assert pol.num_robot_groups == 2 and pol.robot_group_types == ['lifting_robot', 'pushing_robot']
    assert [n.num_output_channels for n in pol.policy_nets] == [2, 1]
    for i, seed in enumerate((51, 52)):
        pol.policy_nets[i].load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(4, [2, 1][i], seed)))
    random.seed(5)       # the reference seeds in __init__, before the nets exist; re-seed after loading
    s = synth.make_states(3, 4, 61)
    state = [[s[0], None], [s[1]]]
    acts = [pol.step(state, exploration_eps=e) for e in (0.0, 0.5, 1.0, 0.5)]
    assert [[a[0][0], a[1][0]] for a in acts] == g['actions'].tolist()
    a, info = pol.step([[None, s[2]], [None]], exploration_eps=0.0, debug=True)
    assert a[0][1] == int(g['debug_action'][0]) and a[1][0] is None and a[0][0] is None
    assert rel(info['output'][0][1], g['debug_output']) < TOL
    t = pol.apply_transform(s[0])
    assert tuple(t.shape) == (1, 4, 96, 96) and torch.equal(t, olearner.apply_transform(s[0]))
    assert pol.build_network.__func__ is pol.build_policy_nets.__func__
    nets = pol.build_policy_nets()
    assert len(nets) == 2 and not torch.equal(nets[0].flat_params, pol.policy_nets[0].flat_params)


def test_device_replay_buffer_matches_host_sampling(simq_mod):
    cin, cout, n, B = 4, 2, 40, 8
    trs = synth.make_transitions(n, cin, cout, 5, terminal_frac=0.2)
    host = simq_mod.ReplayBuffer(32)
    devb = simq_mod.DeviceReplayBuffer(32, cin)
    for t in trs:                      # 40 pushes into capacity 32: wraps
        host.push(*t)
        devb.push(*t)
    assert len(host) == len(devb) == 32 and host.position == devb.position == 8
    random.seed(123)
    hb = host.sample(B)
    random.seed(123)
    db = devb.sample(B)
    assert torch.equal(db.state.cpu(), torch.from_numpy(np.stack(hb.state)))
    assert db.action.cpu().tolist() == list(hb.action)
    assert rel(db.reward, np.asarray(hb.reward, np.float32)) < 1e-7
    assert db.non_final_mask == [s is not None for s in hb.next_state]
    assert torch.equal(db.next_state.cpu(), torch.from_numpy(np.stack([s for s in hb.next_state if s is not None])))
    # identical TD step from either buffer
    cfg = cases.make_cfg(B)
    outs = []
    for b in (hb, db):
        policy, target = make_net(simq_mod, cin, cout, 7, True), make_net(simq_mod, cin, cout, 8, False)
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        outs.append(simq_mod.train(cfg, policy, target, opt, b, None, cases.GAMMA))
    assert abs(outs[0]['loss'] - outs[1]['loss']) <= 1e-6 * abs(outs[0]['loss'])


def test_target_sync_and_checkpoint_format(simq_mod, tmp_path):
    policy, target = make_net(simq_mod, 4, 2, 1, True), make_net(simq_mod, 4, 2, 2, False)
    target.load_state_dict(policy.state_dict())                  # train.py:214,269
    assert torch.equal(target.flat_params, policy.flat_params) and torch.equal(target.bn_buffers, policy.bn_buffers)
    other = make_net(simq_mod, 4, 2, 3, False)
    other.copy_state_from(policy)
    assert torch.equal(other.flat_params, policy.flat_params)
    path = tmp_path / 'policy_00000001.pth.tar'                  # train.py:315-322 format
    torch.save({'timestep': 1, 'state_dicts': [policy.state_dict()]}, str(path))
    ck = torch.load(str(path), map_location='cpu')
    assert len(ck['state_dicts'][0]) == 138 and 'module.resnet18.fc.weight' in ck['state_dicts'][0]
    assert tuple(ck['state_dicts'][0]['module.resnet18.layer1.0.conv1.weight'].shape) == (64, 64, 3, 3)


# ---- matrix-core precisions ---------------------------------------------------------------------------------------
# Opt-in arithmetic modes of the 3x3 / 1x1 convolutions (the default 'fp32' is the exact one held to 1e-4 above):
#   'bf16x3'  split-bf16 (v = hi + lo, 3 MFMA products, fp32 accumulate): ~10x the round-off of true fp32
#             (tests/diag/diag_fwd_error.py: eval Q-map 1.7e-5 vs 2.4e-6).  Bars: Q / loss / td <= 3e-4 (train-mode BN on
#             2-4 samples amplifies it to ~1e-4), gradients <= max(100 x reference-fp32 error, 5e-2) vs fp64.
#   'bf16'    plain bf16 operands AND bf16 pre-BatchNorm convolution outputs (BASELINE configs 3 and 5): the mixed-precision recipe
#             of torch.autocast.  Bars are CALIBRATED (fixture G8, tests/golden/bf16_calibration.npz: the reference's own modules
#             under torch.autocast('cpu', bfloat16) vs the fp64 oracle on these very inputs -- eval Q-maps 1.7-4.7e-2, train-mode
#             Q-maps 4.6-8.7e-2, loss up to 16 %, gradients 0.45-0.54 on the 4-8 sample batches): Q-map <= max(5e-2, 1.5 x
#             calibration), loss / td-error <= max(15 %, 2 x calibration), gradients <= max(0.5, 1.2 x calibration).
@pytest.mark.parametrize('precision,tol', [('bf16x3', 3e-4), ('bf16', 5e-2)])
@pytest.mark.parametrize('case', cases.FORWARD_CASES, ids=[c[0] for c in cases.FORWARD_CASES])
def test_precision_forward(simq_mod, case, precision, tol, golden_dir):
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    x_hwc = torch.from_numpy(synth.make_states(B, cin, dseed)).cuda()
    tol_eval = tol_train = tol
    if precision == 'bf16':
        cal = np.load('%s/bf16_calibration.npz' % golden_dir)
        tol_eval, tol_train = max(tol, 1.5 * float(cal[name + '.eval'])), max(tol, 1.5 * float(cal[name + '.train']))
    net = make_net(simq_mod, cin, cout, wseed, training=False, precision=precision)
    with torch.no_grad():
        q = net.forward_nhwc(x_hwc)
    err_eval = rel(q, g['q_eval'])
    net = make_net(simq_mod, cin, cout, wseed, training=True, precision=precision)
    with torch.no_grad():
        q = net.forward_nhwc(x_hwc)
    err_train = rel(q, g['q_train'])
    print('\n[%s %s] Q-map error eval %.3g (bar %.3g) train-mode %.3g (bar %.3g)' % (name, precision, err_eval, tol_eval, err_train, tol_train))
    assert err_eval < tol_eval and err_train < tol_train
    sd = net.state_dict()
    got = np.concatenate([sd[k].cpu().double().numpy().ravel() for k in sd
                          if k.endswith('running_mean') or k.endswith('running_var')])
    assert rel(got, g['bn_buffers_after']) < tol


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('case', cases.TRAIN_CASES + cases.TRAIN_CASES_CIN, ids=[c[0] for c in cases.TRAIN_CASES + cases.TRAIN_CASES_CIN])
def test_precision_fused_train(simq_mod, case, precision, golden_dir):
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    cfg = cases.make_cfg(B)
    batch = cases.make_batch(cin, cout, B, dseed)
    spec = ofcn.state_spec(cin, cout)
    policy = make_net(simq_mod, cin, cout, wseed, training=True, precision=precision)
    target = make_net(simq_mod, cin, cout, wseed + 1000, training=False, precision=precision)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
    ex64 = {}
    olearner.train_step(cfg, st64, tg64, spec, [None] * len(olearner.grad_keys(spec)), batch, cases.GAMMA, cases.LR,
                        cases.MOMENTUM, cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
    info = simq_mod.train(cfg, policy, target, opt, batch, None, cases.GAMMA)
    ref_err = float(g['ref_fp32_grad_relerr'])
    tn = float(policy._simq_opt_state.total_norm.item())
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    got = {k: v / coef for k, v in grads_to_reference_layout(policy).items()}
    err = global_rel_l2(got, ex64['grads'])
    print('\n[%s %s] loss %.6f (ref %.6f)  grad rel-L2 err vs fp64 %.3g (reference fp32: %.3g)' % (
        name, precision, info['loss'], float(g['loss'][0]), err, ref_err))
    if precision == 'bf16x3':
        assert rel(info['loss'], g['loss'][0]) < 3e-4 and rel(info['td_error'], g['td_error'][0]) < 3e-4
        assert rel(policy._last['q_sa'], g['q_sa']) < 3e-4 and rel(policy._last['y'], g['y']) < 3e-4
        assert err <= max(100 * ref_err, 5e-2)
    else:
        cal = np.load('%s/%s.npz' % (golden_dir, 'bf16_calibration_cin' if case in cases.TRAIN_CASES_CIN else 'bf16_calibration'))
        print('   bf16 calibration (reference under torch.autocast): loss %.3g td %.3g grad %.3g' % (
            float(cal[name + '.loss']), float(cal[name + '.td_error']), float(cal[name + '.grad'])))
        assert rel(info['loss'], g['loss'][0]) < max(0.15, 2 * float(cal[name + '.loss']))
        assert rel(info['td_error'], g['td_error'][0]) < max(0.15, 2 * float(cal[name + '.td_error']))
        assert err <= max(0.5, 1.2 * float(cal[name + '.grad']))
    info2 = simq_mod.train(cfg, policy, target, opt, batch, None, cases.GAMMA)
    assert np.isfinite(info2['loss'])
    sd = policy.state_dict()
    assert all(int(sd[k]) == 4 for k in sd if k.endswith('num_batches_tracked'))


def test_policy_step_batches_the_robots_of_a_group(simq_mod):
    """SURVEY 8f (batched multi-env inference): the robots of one group share a single eval forward; actions, debug
    Q-maps and the RNG draw order are those of the reference's per-robot loop (policies.py:57-66)."""
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 4}, {'pushing_robot': 2}], num_input_channels=4,
                                final_exploration=0.01, checkpoint_path=None)
    pol = simq_mod.DQNPolicy(cfg, train=False, random_seed=3)
    s = synth.make_states(6, 4, 77)
    state = [[s[0], s[1], None, s[2]], [s[3], s[4]]]
    random.seed(11)
    a_batched, info = pol.step(state, exploration_eps=0.3, debug=True)
    # the same thing one robot at a time (every group call sees exactly one live state -> batch-1 path)
    random.seed(11)
    a_single = [[None] * 4, [None] * 2]
    q_single = [[None] * 4, [None] * 2]
    for i, g in enumerate(state):
        for j, st in enumerate(g):
            if st is None:
                continue
            one = [[None] * len(gg) for gg in state]
            one[i][j] = st
            a, inf = pol.step(one, exploration_eps=0.3, debug=True)
            a_single[i][j], q_single[i][j] = a[i][j], inf['output'][i][j]
    assert a_batched == a_single
    for i, g in enumerate(state):
        for j, st in enumerate(g):
            if st is None:
                assert info['output'][i][j] is None
            else:
                assert rel(info['output'][i][j], q_single[i][j]) < 1e-6
    assert pol.step(state, exploration_eps=0.0) == [[int(np.argmax(info['output'][0][j].reshape(-1))) if state[0][j] is not None else None for j in range(4)],
                                                    [int(np.argmax(info['output'][1][j].reshape(-1))) for j in range(2)]]


def test_aliased_device_replay_buffer(simq_mod):
    """observations stored once in HBM.  A collector-style stream (TransitionTracker output, next_state of t
    is the state object of t+1) through AliasedDeviceReplayBuffer vs the reference-style host ReplayBuffer: same picks,
    same minibatch contents, one pool slot per observation, slots recycled when the ring wraps."""
    from simq.learner import AliasedDeviceReplayBuffer, ReplayBuffer, TransitionTracker, assemble_batch
    rng = np.random.RandomState(3)
    C, cap = 4, 24
    obs = lambda: rng.rand(96, 96, C).astype(np.float32)
    tracker = TransitionTracker([[obs(), obs()]])
    dev_buf, host_buf = AliasedDeviceReplayBuffer(cap, C, pool_slots=cap + 8), ReplayBuffer(cap)
    pushed = 0
    for t in range(60):
        tracker.update_action([[int(rng.randint(2 * 96 * 96)), int(rng.randint(2 * 96 * 96))]])
        done = t % 17 == 16
        state = [[obs() if (rng.rand() < 0.8 and not done) else None for _ in range(2)]]
        for tr in tracker.update_step_completed([[float(rng.randn()), float(rng.randn())]], state, done)[0]:
            dev_buf.push(*tr)
            host_buf.push(*tr)
            pushed += 1
        if done:
            tracker = TransitionTracker([[obs(), obs()]])
    assert pushed > cap and len(dev_buf) == len(host_buf) == cap and dev_buf.position == host_buf.position
    # aliasing: at most one slot per live observation (<= cap + robots), far below the 2 * cap of the two-ring layout
    assert dev_buf.observations_resident <= cap + 4
    for seed in (1, 2):
        random.seed(seed)
        hb = host_buf.sample(8)
        random.seed(seed)
        db = dev_buf.sample(8)
        ref = assemble_batch(hb, torch.device('cuda'))
        assert torch.equal(db.state, ref.state) and torch.equal(db.next_state, ref.next_state)
        assert torch.equal(db.action, ref.action) and torch.equal(db.reward, ref.reward)
        assert db.non_final_mask == ref.non_final_mask and torch.equal(db.nonfinal_pos, ref.nonfinal_pos)
    # independent arrays (no aliasing) exhaust a pool sized for aliasing -> loud error, not silent corruption
    small = AliasedDeviceReplayBuffer(8, C, pool_slots=9)
    with pytest.raises(Exception, match="pool exhausted"):
        for _ in range(8):
            small.push(obs(), 0, 0.0, obs())


@pytest.mark.parametrize('double_dqn', [True, False], ids=['double', 'vanilla'])
def test_library_fused_step_equals_composed_step(simq_mod, double_dqn):
    """simq_train_step (one C call) against the same launches issued one by one from Python (the data-parallel form):
    loss / td / q_sa / TD targets identical, parameters after two steps equal up to the atomics' summation order."""
    import simq.learner as sl
    cin, cout, B = 4, 2, 6
    batch = cases.make_batch(cin, cout, B, 123)
    res = []
    for fused in (True, False):
        policy, target = make_net(simq_mod, cin, cout, 61, True), make_net(simq_mod, cin, cout, 62, False)
        infos = [sl.train_step(policy, target, batch, cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP,
                               use_double_dqn=double_dqn, options=sl.StepOptions(fused=fused)) for _ in range(2)]
        res.append((infos, policy._last['q_sa'].clone(), policy._last['y'].clone(), policy.flat_params.clone(),
                    policy.bn_buffers.clone(), dict(policy.num_batches_tracked)))
    (ia, qa, ya, pa, ba, na), (ib, qb, yb, pb, bb, nb) = res
    assert abs(ia[0]['loss'] - ib[0]['loss']) <= 1e-6 * abs(ib[0]['loss']) and abs(ia[0]['td_error'] - ib[0]['td_error']) <= 1e-6 * abs(ib[0]['td_error'])
    assert abs(ia[1]['loss'] - ib[1]['loss']) <= 2e-2 * abs(ib[1]['loss'])        # second step sees the (atomics-ordered) first update
    assert na == nb and all(v == (4 if double_dqn else 2) for v in na.values())
    assert rel(pa, pb) < 1e-3 and rel(ba, bb) < 1e-3


@pytest.mark.parametrize('cout', [2, 1], ids=['cout2', 'cout1'])
def test_onehot_backward_equals_dense_backward(simq_mod, cout):
    """simq_backward_onehot (the head starts from the B non-zero pixels of dLoss/dQ) against simq_backward on the dense
    dQ map that simq_td_huber writes: same gradient up to fp32 summation order; border pixels (bilinear taps that coincide)
    and both output channels are hit."""
    from simq._lib import MODE_TRAIN, lib, ptr, stream_ptr
    cin, B = 4, 6
    net = make_net(simq_mod, cin, cout, 77, True)
    x = torch.from_numpy(synth.make_states(B, cin, 5)).cuda()
    q = net._forward_raw(x, MODE_TRAIN)
    n = cout * 96 * 96
    pix = [0, 95, 96 * 95, 96 * 96 - 1, 96 * 40 + 17, 96 * 3 + 94]                   # corners, edges, interior
    action = torch.tensor([(i % cout) * 9216 + pix[i] for i in range(B)], dtype=torch.int64, device='cuda')
    reward = torch.linspace(-1.5, 2.0, B, device='cuda')
    nsv = torch.linspace(-0.5, 0.5, B, device='cuda')
    outs = [torch.empty(B, device='cuda') for _ in range(3)]
    out4 = torch.empty(4, device='cuda')
    dq = torch.empty_like(q)
    lib.call('simq_td_huber', ptr(q), B, n, ptr(action), ptr(reward), ptr(nsv), 0.75, 1.0 / B, ptr(outs[0]), ptr(outs[1]),
             ptr(outs[2]), ptr(out4), ptr(dq), stream_ptr(torch.device('cuda')))
    assert int((dq != 0).sum()) <= B
    g_dense = net._backward_raw(dq, B).clone()
    g_onehot = net._backward_onehot(action, outs[0], outs[1], 1.0 / B, B).clone()
    assert rel(g_onehot, g_dense) < 1e-4
    num = float((g_onehot.double() - g_dense.double()).norm() / g_dense.double().norm())
    assert num < 1e-4, num


@pytest.mark.parametrize('fixture,case_list,worst', [('grad_study.npz', cases.GRAD_STUDY_CASES, 10.0), ('grad_study_b64.npz', cases.GRAD_STUDY_B64_CASES, 10.0),
                                                     ('grad_study_b32.npz', cases.GRAD_STUDY_B32_CASES, 10.0),
                                                     ('grad_study_b128.npz', cases.GRAD_STUDY_B128_CASES, 3.0)],
                         ids=['b8_b32', 'b64', 'b32x12', 'b128x12'])
def test_gradient_parity_distribution(simq_mod, golden_dir, fixture, case_list, worst):
    """SURVEY section 0's criterion for gradients, err_build <= k * err_reference-fp32, judged as a DISTRIBUTION (fixture
    tests/golden/grad_study.npz, written by oracle/gen_golden.py from the imported reference: 10 seeded B=8 and 3 seeded B=32 batches,
    each with the fp64 oracle's gradient / first-update / second-step-loss and the error the REFERENCE's own fp32 train.train makes on
    the same 16 sampled elements per tensor).  fp32 gradients of these batches are 1e-4 .. 1e-2 accurate for any implementation and
    which one is luckier changes per batch, so: median HIP error <= 2 x median reference error, no case beyond 10 x the reference's
    error on that case (or its median), for the pre-clip gradient and for the first parameter update (sampled elements, not
    tensor norms); the second call is checked per transition against the fp64 oracle run from the post-step-1 state.
    Measured (MI355X): medians 1.6e-3 .. 2.3e-3 (HIP, run to run) vs 2.0e-3 (reference fp32) for both gradient and update; B = 64: 3.7e-3 vs 2.7e-3.
    Second fixture (grad_study_b64.npz, six seeded batches of 64 = configs[3]'s per-GPU batch, where the fp32 plans pick their large-batch
    tiles and 36-plane Winograd problems): the same bars.  Third fixture (grad_study_b32.npz, twelve seeded batches of the headline workload's
    own shape -- 32 transitions, Cin 4, Cout 2): the same bars; it is the sample the choice of F(4x4,3x3) for the grad-mode forward of
    layer4's 512->512 convolutions rests on (tests/diag_f4_grad_layers.py, docs/history.md 4).  Fourth fixture (grad_study_b128.npz, round 4: twelve
    batches of 128 = configs[2] / configs[4]'s per-GPU shape, Cin 5, Cout 2; reference fp32 median 2.9e-3): median <= 2 x (measured 1.5 x), eleven of
    the twelve cases within 3 x (measured: within 2 x) and none beyond 10 x.  The twelfth, gs_b128_09, sits at 6.8 x (2.0e-2) whatever the plan
    computes with -- Winograd forms or direct convolutions, fused or separate BatchNorm sums, one-hot or dense backward all give 2.01e-2 .. 2.06e-2 --
    and tests/diag/diag_b128_maskflip.py shows why: of the 16 256 ReLU-mask elements of the head's last activation that the one-hot TD gradient
    enters the network through, exactly ONE differs from the fp64 oracle's (transition 55, pixel (39, 7), channel 17: pre-activation -1.5e-6 in
    fp64, +1.5e-5 after this forward's fp32 round-off); the train-mode BatchNorm backward behind it, which cancels the one-hot gradient against
    its dense mean terms, turns that 6e-5 share of the signal into 2 % of the gradient.  Any fp32 forward pays this on some batch (the
    reference's own worst case of the twelve is 8.3e-3 on gs_b128_01, where the HIP path measures 5.2e-3); it is a property of the loss (a
    one-hot gradient through a ReLU), not of a kernel."""
    from oracle import learner as olearner
    g = np.load('%s/%s' % (golden_dir, fixture))
    rows = []
    for name, cin, cout, B, wseed, dseed in case_list:
        cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
        policy, target = make_net(simq_mod, cin, cout, wseed, True), make_net(simq_mod, cin, cout, wseed + 1000, False)
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        # (round 6, suite budget: the 16 sampled elements per tensor are picked ON the device -- three host copies of 11 M doubles per case were most
        # of this test's time)
        samp = [torch.tensor(cases.sample_indices(v.numel()), device='cuda') for v in policy.reference_views(policy.flat_params)]
        pick = lambda flat: [v.detach().reshape(-1)[i].double().cpu() for v, i in zip(policy.reference_views(flat), samp)]
        p0 = pick(policy.flat_params)
        info1 = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
        tn = float(policy._simq_opt_state.total_norm.item())
        coef = min(1.0, cases.CLIP / (tn + 1e-6))
        grads = [v / coef for v in pick(policy.flat_grads)]
        p1 = pick(policy.flat_params)
        sd1, sd_target = (step2_oracle.snapshot(policy), step2_oracle.snapshot(target)) if len(rows) < (2 if B >= 64 else 3) or (B <= 8 and len(rows) < 6) else (None, None)       # (the fp64 second-step oracle costs 3-8 s of host time per case at B >= 32)
        info2 = simq_mod.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
        if sd1 is not None:     # the second call per transition against the fp64 oracle run from the post-step-1 state (tests/step2_oracle.py)
            step2_oracle.second_step_against_the_oracle(sd1, sd_target, batch, policy._last['q_sa'].cpu().numpy(), policy._last['y'].cpu().numpy(), info2)
        gs, ds = [], []
        for t, a, b in zip(grads, p0, p1):
            gs.append(t.numpy())
            ds.append((b - a).numpy())
        rl2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
        rows.append(dict(name=name, grad=rl2(np.stack(gs), g[name + '.grad64']), ref_grad=float(g[name + '.ref_grad_err']),
                         dparam=rl2(np.stack(ds), g[name + '.dparam64']), ref_dparam=float(g[name + '.ref_dparam_err']),
                         loss1=abs(info1['loss'] - float(g[name + '.loss64'][0])) / float(g[name + '.loss64'][0]),
                         loss2=abs(info2['loss'] - float(g[name + '.loss64'][1])) / float(g[name + '.loss64'][1]),
                         ref_loss2=float(g[name + '.ref_loss_err'][1]),
                         norm=abs(tn - float(g[name + '.total_norm64'])) / float(g[name + '.total_norm64'])))
        del policy, target, opt
    print()
    for r in rows:
        print('%-10s grad %.3g (ref %.3g)  update %.3g (ref %.3g)  loss step 1 %.2g  step 2 %.3g (ref %.3g)  |g| %.2g'
              % (r['name'], r['grad'], r['ref_grad'], r['dparam'], r['ref_dparam'], r['loss1'], r['loss2'], r['ref_loss2'], r['norm']))
    med = lambda k: float(np.median([r[k] for r in rows]))
    print('medians: grad %.3g (ref %.3g)  update %.3g (ref %.3g)  loss step 2 %.3g (ref %.3g)'
          % (med('grad'), med('ref_grad'), med('dparam'), med('ref_dparam'), med('loss2'), med('ref_loss2')))
    assert all(r['loss1'] < 1e-4 for r in rows)                                    # the forward / Huber side: the 1e-4 bar
    for k, rk in (('grad', 'ref_grad'), ('dparam', 'ref_dparam')):
        assert med(k) <= 2.0 * med(rk), (k, med(k), med(rk))
        ratios = sorted(r[k] / max(r[rk], med(rk)) for r in rows)
        assert ratios[-1] <= 10.0, (k, ratios)
        if worst < 10.0:          # the B = 128 study: at most ONE case beyond `worst` x (a ReLU-mask flip under the one-hot gradient, see the docstring)
            assert ratios[-2] <= worst, (k, ratios)
    # The second step's loss against the fp64 TRAJECTORY is the first update's error seen through a chaotic synthetic problem: in the
    # fixture the reference's own fp32 second-step loss is off by up to 15 x its update error (0.10 on gs_b8_03), with a heavy tail
    # (13 samples: 2e-5 .. 1e-1), and one summation order against another moves it by several per cent (tests/diag_step2_sensitivity.py).
    # What "the second call is right" means is checked above per transition against the oracle (every B = 8 case and the first six of
    # each fixture, three at B = 128); the trajectory keeps a distribution bar -- median within 4 x the reference's median, no case beyond 25 % --
    # AND a reference-relative bar per case: the reference's own fp32 turns its first-update error into a second-step loss error with some
    # amplification (ref_loss2 / ref_dparam); no HIP case may exceed that amplification applied to its OWN update error, with a floor of
    # the reference's median second-step error.
    assert med('loss2') <= 4.0 * med('ref_loss2'), (med('loss2'), med('ref_loss2'))
    assert all(r['loss2'] < 0.25 for r in rows), [r['loss2'] for r in rows]
    # (amplification = second-step loss error / first-update error: 2 .. 80 for the reference's own fp32 over the four fixtures, 62 for the
    # HIP path's unluckiest case -- a chaotic quantity, so the bar is the LARGEST the reference shows anywhere with a margin, not this
    # fixture's own maximum; what it excludes is a second step that is wrong by more than the first update's error can explain)
    amp = max([100.0] + [r['ref_loss2'] / max(r['ref_dparam'], 1e-12) for r in rows])
    for r in rows:
        assert r['loss2'] <= amp * r['dparam'] + 4.0 * med('ref_loss2') + 1e-4, ('second-step loss beyond what the first update\'s error explains', r, amp)


def test_optimizer_state_is_interchangeable_with_the_reference_layout(simq_mod, tmp_path):
    """train.py:204 / :331 -- optimizer.load_state_dict / optimizer.state_dict() in a checkpoint.  The reference's momentum buffers
    are OIHW tensors; simq stores OHWI and presents every parameter / gradient / momentum buffer to torch under the reference's
    logical [O,I,H,W] shape (a permuted view).  (1) A state dict as the reference's SGD writes it after >= 1 step (OIHW buffers,
    here seeded random values, pickled through torch.save) is adopted element by logical index and the NEXT update is the
    reference's sgd_step arithmetic on it; (2) a simq-written optimizer state loads into a plain torch SGD over OIHW-contiguous
    reference-shaped parameters and steps there with identical logical buffers (Cin = 7: the stem weight [64,7,7,7] has the
    same shape in both layouts, so a shape-based guess could not tell them apart)."""
    import os
    from oracle import learner as olearner
    cin, cout, B = 7, 2, 3
    spec = ofcn.state_spec(cin, cout)
    ref_shapes = [tuple(s) for k, s, kind in spec if ofcn.has_gradient(kind)]
    policy, target = make_net(simq_mod, cin, cout, 81, True), make_net(simq_mod, cin, cout, 82, False)
    params = list(policy.parameters())
    trainable = [p for p in params if tuple(p.shape) != (1000, 512) and tuple(p.shape) != (1000,)]
    assert [tuple(p.shape) for p in trainable] == ref_shapes                     # parameters carry the reference's shapes
    # (1) reference-written state: one OIHW momentum buffer per parameter that received a gradient, indexed like torch does
    g = torch.Generator().manual_seed(5)
    ref_bufs = [torch.randn(s, generator=g) * 1e-3 for s in ref_shapes]
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    sd = opt.state_dict()
    index_of = {id(p): i for i, p in enumerate(params)}
    sd['state'] = {index_of[id(p)]: {'momentum_buffer': b.clone()} for p, b in zip(trainable, ref_bufs)}
    path = os.path.join(str(tmp_path), 'opt.pth.tar')
    torch.save({'optimizers': [sd]}, path)
    opt.load_state_dict(torch.load(path, weights_only=False)['optimizers'][0])    # train.py:200-204
    p_before = [p.detach().clone().cpu() for p in trainable]
    batch = cases.make_batch(cin, cout, B, 83)
    simq_mod.train(cases.make_cfg(B), policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    grads = [v.detach().clone().cpu() for v in policy.reference_views(policy.flat_grads)]     # clipped gradient, logical OIHW
    want_p, want_m = [p.clone() for p in p_before], [b.clone() for b in ref_bufs]
    olearner.sgd_step(want_p, grads, want_m, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY)     # torch.optim.SGD.step restated
    for p, wp, wm in zip(trainable, want_p, want_m):
        got_m = opt.state[p]['momentum_buffer']
        assert tuple(got_m.shape) == tuple(wm.shape)
        assert rel(got_m, wm) < 1e-6 and rel(p, wp) < 1e-6
    # (2) simq-written state -> the reference's optimizer (plain contiguous OIHW parameters on the CPU)
    torch.save({'optimizers': [opt.state_dict()]}, path)
    sd2 = torch.load(path, map_location='cpu', weights_only=False)['optimizers'][0]
    ref_params = [torch.nn.Parameter(p.detach().cpu().contiguous()) for p in params]
    ref_opt = torch.optim.SGD(ref_params, lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    ref_opt.load_state_dict(sd2)
    for rp, p in zip(ref_params, params):
        if id(p) in {id(t) for t in trainable}:
            rp.grad = torch.zeros_like(rp)
            assert torch.equal(ref_opt.state[rp]['momentum_buffer'], opt.state[p]['momentum_buffer'].cpu())
    ref_opt.step()                                                                # shapes agree: the reference can continue
    # a buffer of the wrong shape is refused, not reinterpreted
    bad = torch.optim.SGD(policy.parameters(), lr=0.01, momentum=0.9)
    bad.state[trainable[2]]['momentum_buffer'] = torch.zeros(3, 3, device='cuda')
    from simq._lib import SimqError
    with pytest.raises(SimqError):
        simq_mod.train(cases.make_cfg(B), policy, target, bad, batch, olearner.apply_transform, cases.GAMMA)


def test_checkpoint_files_reference_ring_and_resume(simq_mod, tmp_path, golden_dir):
    """SURVEY 8f row 4 (train.py:197-209, 309-346).  (1) a checkpoint pickled by the reference's own classes goes into the
    HBM ring and samples what the host ring samples; (2) save_policy + save_checkpoint + resume() in fresh objects continue
    the run: same picks, same losses, same weights as the uninterrupted trainer (up to the atomics' summation order)."""
    import os
    from simq import checkpoint as ck
    from simq.learner import AliasedDeviceReplayBuffer, TransitionTracker, assemble_batch
    dev = torch.device('cuda')
    host = ck.load_checkpoint(os.path.join(golden_dir, 'ref_checkpoint.pth.tar'))['replay_buffers'][0]
    for aliased in (True, False):
        ring = ck.to_device_ring(host, aliased=aliased)
        assert ring.position == host.position == 1 and len(ring) == len(host) == 3 and ring.capacity == 3
        if aliased:
            assert ring.observations_resident == 4            # o1, o2 (shared by two records), o3, o4
        random.seed(4)
        db = ring.sample(3)
        random.seed(4)
        ref = assemble_batch(host.sample(3), dev)
        assert torch.equal(db.state, ref.state) and torch.equal(db.next_state, ref.next_state)
        assert torch.equal(db.action, ref.action) and torch.equal(db.reward, ref.reward) and db.non_final_mask == ref.non_final_mask
        back = ck.to_host_ring(ring)
        assert back.position == 1 and all(np.array_equal(a.state, b.state) and a.action == b.action and a.reward == b.reward
                                          for a, b in zip(back.buffer, host.buffer))
        if aliased:
            assert back.buffer[1].next_state is back.buffer[2].state

    cin, cout, B = 4, 2, 6
    cfg = types.SimpleNamespace(batch_size=B, use_double_dqn=True, grad_norm_clipping=100)
    sgd = lambda net: torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    policy, target = make_net(simq_mod, cin, cout, 71, True), make_net(simq_mod, cin, cout, 71, False)
    opt = sgd(policy)
    rng = np.random.RandomState(8)
    obs = lambda: rng.rand(96, 96, cin).astype(np.float32)
    ring = AliasedDeviceReplayBuffer(16, cin)
    tracker = TransitionTracker([[obs()]])
    for t in range(22):                                         # wraps the 16-slot ring
        tracker.update_action([[int(rng.randint(cout * 96 * 96))]])
        done = t % 9 == 8
        for tr in tracker.update_step_completed([[float(rng.randn())]], [[None if done else obs()]], done)[0]:
            ring.push(*tr)
        if done:
            tracker = TransitionTracker([[obs()]])
    random.seed(9)
    for step in (1, 2):
        simq_mod.train(cfg, policy, target, opt, ring.sample(B), None, 0.75)
        ppath = ck.save_policy(tmp_path, step, [policy])
        cpath = ck.save_checkpoint(tmp_path, step, 0, [opt], [ring])
    assert sorted(os.path.basename(p) for p in tmp_path.iterdir()) == \
        ['checkpoint_00000002.pth.tar', 'policy_00000001.pth.tar', 'policy_00000002.pth.tar']   # train.py:341-345
    target.load_state_dict(policy.state_dict())                 # what a resumed run starts from (train.py:212-214)
    rstate = random.getstate()
    cont = [simq_mod.train(cfg, policy, target, opt, ring.sample(B), None, 0.75)]
    p_cont1 = policy.flat_params.clone()
    cont.append(simq_mod.train(cfg, policy, target, opt, ring.sample(B), None, 0.75))

    policy2, target2 = simq_mod.FCN(cin, cout), simq_mod.FCN(cin, cout)
    policy2.load_state_dict(torch.load(ppath, map_location=dev)['state_dicts'][0])
    policy2.train()
    target2.load_state_dict(policy2.state_dict())
    target2.eval()
    opt2 = sgd(policy2)
    start, episode, rings = ck.resume(cpath, [opt2])
    assert (start, episode) == (2, 0) and isinstance(rings[0], AliasedDeviceReplayBuffer)
    assert rings[0].position == ring.position and len(rings[0]) == len(ring)
    assert rings[0].observations_resident == ring.observations_resident
    random.setstate(rstate)
    again = [simq_mod.train(cfg, policy2, target2, opt2, rings[0].sample(B), None, 0.75)]
    # the first resumed step sees bit-identical parameters, optimizer state and minibatch: only the summation order of the atomically
    # accumulated BN statistics / weight gradients differs (1e-6 on the loss, 1e-5 on the parameters)
    assert abs(again[0]['loss'] - cont[0]['loss']) <= 1e-4 * abs(cont[0]['loss'])
    assert abs(again[0]['td_error'] - cont[0]['td_error']) <= 1e-4 * abs(cont[0]['td_error'])
    assert rel(policy2.flat_params, p_cont1) < 1e-4
    # the second step sees that noise through the network once more (this file's header: a double-DQN argmax of one of the 6
    # transitions may even flip): same minibatch, loss of the same size, parameters and momentum (= the last gradients) close
    again.append(simq_mod.train(cfg, policy2, target2, opt2, rings[0].sample(B), None, 0.75))
    assert np.isfinite(again[1]['loss']) and abs(again[1]['loss'] - cont[1]['loss']) <= 0.2 * abs(cont[1]['loss'])
    assert rel(policy2.flat_params, policy.flat_params) < 1e-2
    assert rel(policy2._simq_opt_state.momentum, policy._simq_opt_state.momentum) < 0.5


def test_winograd_layers_equal_direct_convolution_network(simq_mod):
    """The fp32 plan runs its 128- to 512-channel 3x3 layers in Winograd form (conv_winograd.hip).  A plan created with
    simq_plan_options.winograd = 0 (every layer on the implicit-GEMM kernel) gives the same Q-maps, batch statistics, loss and
    gradients to fp32 round-off; the options a plan was created with are reported back by simq_plan_get_options."""
    import simq.learner as sl
    cin, cout, B = 4, 2, 6
    batch = cases.make_batch(cin, cout, B, 321)
    res = []
    for on in (1, 0):
        opts = {'winograd': on}
        policy, target = make_net(simq_mod, cin, cout, 81, True, options=opts), make_net(simq_mod, cin, cout, 82, False, options=opts)
        assert policy.plan.options['winograd'] == on and policy.plan.options['winograd_f4_grad'] == 2
        info = sl.train_step(policy, target, batch, cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP,
                             use_double_dqn=True)
        res.append((info, policy._last['q'].clone(), policy._last['y'].clone(), policy.flat_grads.clone(), policy.bn_buffers.clone()))
    (ia, qa, ya, ga, ba), (ib, qb, yb, gb, bb) = res
    assert rel(qa, qb) < 1e-5 and rel(ya, yb) < 1e-5 and rel(ba, bb) < 1e-5
    assert abs(ia['loss'] - ib['loss']) <= 1e-5 * abs(ib['loss'])
    # gradient conditioning, see this file's header (both forms are judged against the fp64 oracle elsewhere)
    assert float((ga - gb).double().norm() / gb.double().norm()) < 2e-2
    with pytest.raises(Exception):
        simq_mod.FCN(cin, cout, options={'no_such_option': 1})
    with pytest.raises(Exception):
        simq_mod.FCN(cin, cout, options={'winograd_f4_grad': 7})


def test_replay_push_stages_through_pinned_memory(simq_mod):
    """Collector hand-off: push() copies the observation into a pinned staging slot and issues an asynchronous
    H2D copy.  The caller may overwrite its ndarray right after push() returns, and more pushes than staging slots recycle
    the slots without corrupting earlier uploads."""
    from simq.learner import AliasedDeviceReplayBuffer, DeviceReplayBuffer
    rng = np.random.RandomState(12)
    C = 4
    for cls in (DeviceReplayBuffer, AliasedDeviceReplayBuffer):
        ring = cls(80, C)
        assert ring._staging.enabled and ring._staging.buf.is_pinned()
        kept, scratch = [], np.empty((96, 96, C), dtype=np.float32)
        for i in range(70):                                    # > 2 x 32 staging slots
            obs = rng.rand(96, 96, C).astype(np.float32)
            nxt = rng.rand(96, 96, C).astype(np.float32)
            kept.append((obs.copy(), nxt.copy()))
            if cls is DeviceReplayBuffer:
                scratch[...] = obs
                ring.push(scratch, i, 0.5, nxt)
                scratch[...] = -1.0                            # the collector reuses its buffer immediately
            else:                                              # the aliased ring identifies an observation by its ndarray OBJECT
                ring.push(obs, i, 0.5, nxt)                    # (train.py:61-66), so every observation is its own array ...
                obs[...] = -1.0                                # ... which may still be overwritten once push() has returned
            nxt[...] = -2.0
        torch.cuda.synchronize()
        for i in (0, 1, 31, 32, 33, 64, 69):
            rec = ring.buffer[i]
            assert np.array_equal(np.asarray(rec.state), kept[i][0]) and np.array_equal(np.asarray(rec.next_state), kept[i][1])


def test_bf16_backward_agrees_across_its_storage_switches(simq_mod):
    """Plain-bf16 plans keep the activation gradients between the residual blocks' kernels in bf16, consume ReLU masks / residuals as
    bf16 planes and fuse the BatchNorm-backward sums into the dgrad epilogues (plan.h: Ctx::gbf, planes_only; backward.hip: fuse_block_out).  Each
    is a simq_plan_options field whose other setting falls back to the generic kernels (bf16_act_grads = 0: fp32 gradients,
    keep_fp32_activations = 1: fp32 activation copies, fuse_bn_backward_sums = 0: separate reduction launches).  One seeded train step
    in every combination that changes the kernels taken; variants that share a forward must give the same loss and gradients that
    differ only by where a gradient is rounded."""
    cin, cout, B = 4, 2, 8
    variants = {'default': {}, 'fp32_grads': {'bf16_act_grads': 0}, 'fp32_act': {'keep_fp32_activations': 1},
                'no_fuse': {'fuse_bn_backward_sums': 0}, 'fp32_both': {'bf16_act_grads': 0, 'keep_fp32_activations': 1}}
    out = {}
    for name, opts in variants.items():
        policy = make_net(simq_mod, cin, cout, 3, True, precision='bf16', options=opts)
        target = make_net(simq_mod, cin, cout, 1003, False, precision='bf16', options=opts)
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        info = simq_mod.train(cases.make_cfg(B), policy, target, opt, cases.make_batch(cin, cout, B, 11), olearner.apply_transform, cases.GAMMA)
        out[name] = {'grads': policy.flat_grads.detach().cpu().numpy(), 'loss': np.float64(info['loss'])}
        del policy, target, opt
    rel = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
    G = {k: v['grads'].astype(np.float64) for k, v in out.items()}
    L = {k: float(v['loss']) for k, v in out.items()}
    print()
    for name in variants:
        print('bf16 backward variant %-10s loss %.9g  gradient vs default %.3g  vs fp32 activation copies + fp32 gradients %.3g'
              % (name, L[name], rel(G[name], G['default']), rel(G[name], G['fp32_both'])))
        assert np.isfinite(G[name]).all()
    # same forward (planes only / fp32 copies kept): identical loss, and the backward differs only by where a gradient is rounded
    same = lambda a, b: abs(a - b) <= 1e-6 * abs(b)                 # (fp64 atomics: the summation order of the BN statistics may differ)
    assert same(L['default'], L['fp32_grads']) and same(L['fp32_act'], L['fp32_both']) and same(L['no_fuse'], L['fp32_both'])
    assert rel(G['default'], G['fp32_grads']) < 3e-2
    assert rel(G['fp32_act'], G['fp32_both']) < 3e-2
    assert rel(G['no_fuse'], G['fp32_act']) < 3e-2            # separate reduction launches sum the STORED (bf16) gradient, the fused
                                                              # epilogue the fp32 value it is about to round
    # different forward roundings (residuals from fp32 copies instead of bf16 planes) are only printed: the gradient of a bf16 network
    # moves by tens of percent under ANY re-rounding of its forward (fixture G8: the reference's own autocast gradients are 0.2-0.4 off
    # its fp32 ones) and the double-DQN argmax of one of the 8 transitions may flip (a 1/8 step of the loss); the calibrated bars of
    # test_precision_fused_train are what holds the bf16 path to the reference
