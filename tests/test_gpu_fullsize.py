"""GPU: BASELINE.json's full sizes (configs[1]: B=32 fp32; configs[2]: B=128 Cin=5 bf16; configs[3]: two heterogeneous
heads) through size-independent properties -- the CPU oracle would need minutes per step at these sizes:

  * minibatch permutation invariance of loss / td-error / gradient (only the summation order changes),
  * linearity of the backward pass in the upstream gradient,
  * exact invariants of train-mode BatchNorm backward (sum_rows dy = 0  =>  the head-conv bias gradients vanish),
  * clip_grad_norm_ post-condition (||g|| <= max_norm) and SGD first-step identity (momentum buffer == clipped g + wd*p),
  * bit-reproducible eval forward / replay gather, first-index argmax on a constant Q-map,
  * the per-group loop of train.py:255-257 on a lifting (Cout=2) + pushing (Cout=1) policy.
"""
import random
import types

import numpy as np
import pytest
import torch

from oracle import cases
from oracle import fcn as ofcn
from simq import synth

import step2_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def simq_mod():
    import simq
    from simq import _lib  # noqa: F401
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return simq


def make_pair(simq_mod, cin, cout, seed, precision='fp32'):
    policy, target = simq_mod.FCN(cin, cout, precision=precision), simq_mod.FCN(cin, cout, precision=precision)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed)))
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed + 1)))
    policy.train()
    target.eval()
    return policy, target


def relnorm(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def run_step(simq_mod, cin, cout, transitions, B, seed, precision='fp32', clip=cases.CLIP):
    from simq.learner import Transition, train_step
    policy, target = make_pair(simq_mod, cin, cout, seed, precision)
    batch = Transition(*zip(*transitions))
    info = train_step(policy, target, batch, cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, clip)
    return policy, info


def test_b32_permutation_invariance_and_optimizer_invariants(simq_mod):
    cin, cout, B = 4, 2, 32
    trs = synth.make_transitions(B, cin, cout, 17, terminal_frac=0.1)
    p0 = ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 5))
    pol_a, info_a = run_step(simq_mod, cin, cout, trs, B, 5)
    perm = list(range(B))
    random.Random(3).shuffle(perm)
    pol_b, info_b = run_step(simq_mod, cin, cout, [trs[i] for i in perm], B, 5)
    assert abs(info_a['loss'] - info_b['loss']) <= 1e-5 * abs(info_a['loss'])
    assert abs(info_a['td_error'] - info_b['td_error']) <= 1e-5 * abs(info_a['td_error'])
    # same multiset of samples -> same gradient up to summation order (and the ill-conditioning documented in DESIGN 2)
    assert relnorm(pol_a.flat_grads, pol_b.flat_grads) < 2e-2
    assert relnorm(pol_a._last['q_sa'][perm], pol_b._last['q_sa']) < 1e-5
    # clip post-condition and first SGD step identity: m = c*g + wd*p0 ; p1 = p0 - lr*m
    tn = float(pol_a._simq_opt_state.total_norm.item())
    assert tn > cases.CLIP                                   # clipping is active on this workload
    gn = float(pol_a.flat_grads.double().norm())
    assert abs(gn - cases.CLIP) <= 1e-4 * cases.CLIP
    ref = make_pair(simq_mod, cin, cout, 5)[0]
    m = pol_a._simq_opt_state.momentum
    assert relnorm(m, pol_a.flat_grads + cases.WEIGHT_DECAY * ref.flat_params) < 1e-6
    assert relnorm(pol_a.flat_params, ref.flat_params - cases.LR * m) < 1e-6
    # BN-backward invariant: the conv biases that feed a train-mode BN get a (numerically) zero gradient
    for (name, _, _), (off, n, _) in zip(pol_a._param_names, pol_a._grad_views):
        if name in ('conv1.bias', 'conv2.bias'):
            assert float(pol_a.flat_grads[off:off + n].abs().max()) < 1e-5
    del p0


def test_b32_backward_is_linear_in_upstream_gradient(simq_mod):
    cin, cout, B = 4, 2, 32
    policy, _ = make_pair(simq_mod, cin, cout, 9)
    x = torch.from_numpy(synth.make_states(B, cin, 4)).cuda()
    g = torch.Generator(device='cpu').manual_seed(1)
    dq1 = (torch.randn(B, cout, 96, 96, generator=g) * 1e-3).cuda()
    dq2 = (torch.randn(B, cout, 96, 96, generator=g) * 1e-3).cuda()
    from simq._lib import MODE_TRAIN
    policy._forward_raw(x, MODE_TRAIN)
    g1 = policy._backward_raw(dq1, B).clone()
    g2 = policy._backward_raw(dq2, B).clone()
    g12 = policy._backward_raw(2.0 * dq1 - 3.0 * dq2, B).clone()
    assert relnorm(g12, 2.0 * g1 - 3.0 * g2) < 1e-4


def test_b32_eval_forward_reproducible_and_argmax(simq_mod):
    cin, cout, B = 4, 2, 32
    policy, _ = make_pair(simq_mod, cin, cout, 21)
    policy.eval()
    x = torch.from_numpy(synth.make_states(B, cin, 8)).cuda()
    with torch.no_grad():
        q1 = policy.forward_nhwc(x)
        q2 = policy.forward_nhwc(x)
        q_first = policy.forward_nhwc(x[:1].contiguous())
    assert torch.equal(q1, q2)                                            # no atomics on the forward path
    assert relnorm(q_first[0], q1[0]) < 1e-5                              # eval BN: samples are independent
    assert policy.argmax(torch.zeros_like(q1[0])) == 0                    # ties -> first index (policies.py:64)
    flat = q1[3].reshape(-1)
    assert policy.argmax(q1[3]) == int(flat.argmax())


def test_b128_bf16_config_runs_and_is_consistent(simq_mod):
    """configs[2]: lifting_4-small_divider (Cin=5), batch 128, bf16 operands."""
    cin, cout, B = 5, 2, 128
    trs = synth.make_transitions(B, cin, cout, 31, terminal_frac=0.1)
    pol_a, info_a = run_step(simq_mod, cin, cout, trs, B, 13, precision='bf16')
    pol_f, info_f = run_step(simq_mod, cin, cout, trs, B, 13, precision='fp32')
    # Bars from tests/diag/diag_b128_bf16_loss.py (round 4, six seed pairs, this shape): bf16-vs-fp32 loss / TD error 0.2-5.0 %, Q of the
    # taken actions 3.6-9.9 % of their range -- and the summation order alone (half-map vs whole-map kernels on the 128-channel layers,
    # same operands) moves THIS batch's loss from 2.7 % to 5.0 % off fp32: train-mode bf16 storage flips roundings of stored activations
    # and the batch statistics carry them to every pixel.  The tight bf16 statement is the teacher-forced, elementwise
    # bit-exact comparison with the rounded-operand oracle (tests/test_gpu_bf16_points.py); this test only says "same model, finite".
    assert np.isfinite(info_a['loss']) and abs(info_a['loss'] - info_f['loss']) <= 1e-1 * abs(info_f['loss'])
    assert abs(info_a['td_error'] - info_f['td_error']) <= 1e-1 * abs(info_f['td_error'])
    # bf16 Q-values of the taken actions vs the exact-fp32 path on the same batch
    assert float((pol_a._last['q_sa'] - pol_f._last['q_sa']).abs().max() / pol_f._last['q_sa'].abs().max()) < 2e-1
    sd = pol_a.state_dict()
    assert all(int(sd[k]) == 2 for k in sd if k.endswith('num_batches_tracked'))
    assert all(torch.isfinite(v).all() for k, v in sd.items() if v.dtype.is_floating_point)


def test_two_heterogeneous_heads_like_train_py_255(simq_mod):
    """configs[3]: lifting_2_pushing_2 -- one net per robot group (Cout 2 and 1), trained in turn (train.py:255-257)."""
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 2}], num_input_channels=5,
                                final_exploration=0.01, checkpoint_path=None, batch_size=16, use_double_dqn=True,
                                grad_norm_clipping=100, discount_factors=[0.85, 0.85])
    policy = simq_mod.DQNPolicy(cfg, train=True, random_seed=1)
    targets = policy.build_policy_nets()
    for i in range(policy.num_robot_groups):
        targets[i].load_state_dict(policy.policy_nets[i].state_dict())      # train.py:213-216
        targets[i].eval()
        policy.policy_nets[i].train()
    opts = [torch.optim.SGD(n.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4) for n in policy.policy_nets]
    buffers = [simq_mod.ReplayBuffer(64), simq_mod.DeviceReplayBuffer(64, 5)]   # host ring and HBM ring side by side
    for i, cout in enumerate((2, 1)):
        for t in synth.make_transitions(40, 5, cout, 50 + i, terminal_frac=0.1):
            buffers[i].push(*t)
    random.seed(7)
    infos = []
    for step in range(2):
        for i in range(policy.num_robot_groups):                                # train.py:255-257
            batch = buffers[i].sample(cfg.batch_size)
            infos.append(simq_mod.train(cfg, policy.policy_nets[i], targets[i], opts[i], batch, policy.apply_transform,
                                        cfg.discount_factors[i]))
    assert len(infos) == 4 and all(np.isfinite(v['loss']) and np.isfinite(v['td_error']) for v in infos)
    assert [n.num_output_channels for n in policy.policy_nets] == [2, 1]
    # each group's action space matches its head (envs.py:374-376)
    s = synth.make_states(2, 5, 77)
    acts = policy.step([[s[0], None], [None, s[1]]], exploration_eps=0.0)
    assert 0 <= acts[0][0] < 2 * 96 * 96 and 0 <= acts[1][1] < 96 * 96 and acts[0][1] is None and acts[1][0] is None


def test_ragged_and_degenerate_batches(simq_mod):
    """B=1, all-but-one terminal, and the reference's own failure mode (no non-final next state, train.py:112)."""
    from simq.learner import Transition, train_step
    cin, cout = 4, 2
    policy, target = make_pair(simq_mod, cin, cout, 3)
    trs = synth.make_transitions(3, cin, cout, 23, terminal_frac=0.0)
    one = Transition(*zip(*trs[:1]))
    info = train_step(policy, target, one, cases.GAMMA, 1, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP)
    assert np.isfinite(info['loss'])
    mostly_terminal = [(s, a, r, None) for (s, a, r, _) in trs[:2]] + [trs[2]]
    info = train_step(policy, target, Transition(*zip(*mostly_terminal)), cases.GAMMA, 3, cases.LR, cases.MOMENTUM,
                      cases.WEIGHT_DECAY, cases.CLIP)
    assert np.isfinite(info['loss'])
    all_terminal = [(s, a, r, None) for (s, a, r, _) in trs]
    with pytest.raises(simq_mod._lib.SimqError if hasattr(simq_mod, '_lib') else Exception):
        train_step(policy, target, Transition(*zip(*all_terminal)), cases.GAMMA, 3, cases.LR, cases.MOMENTUM,
                   cases.WEIGHT_DECAY, cases.CLIP)
    with pytest.raises(Exception):
        policy.forward_nhwc(torch.zeros(1, 96, 96, cin + 1, device='cuda'))      # wrong channel count


_FP32_FAMILIES_B32 = ('winograd_f4', 'winograd_f2', 'winograd_f4_wgrad', 'stem_conv_f32')


@pytest.mark.parametrize('through_ring', [False, True], ids=['host_batch', 'hbm_ring_early_stream'])
def test_b32_train_step_against_reference_pinned_golden(simq_mod, golden_dir, through_ring):
    """BASELINE configs[1] at its own size: two consecutive simq.train calls on the seeded B=32 batch against
    tests/golden/train_c4o2_b32.npz (written by oracle/gen_golden.py after a bit-exact match of the oracle with the
    imported reference train.train; fp64 gradient summary = per-tensor L2 norm + 16 sampled elements).
    through_ring (round 6): the SCHEDULE bench.py times -- the same 32 transitions pushed into a DeviceReplayBuffer and gathered on the
    upload stream (DeviceBatch.ready_event), so that the second call's target-net forward runs on the early stream beside the first call's
    backward pass and SGD (no host synchronisation between the two calls: the post-step-1 state the fp64 oracle starts from is cloned
    ON the launch stream).  A host batch (the other case) has no ready_event and never takes that path."""
    from oracle import learner as olearner
    from simq import arch
    name, cin, cout, B, wseed, dseed = cases.TRAIN_CASES_FULL[0]
    g = np.load('%s/%s.npz' % (golden_dir, name))
    cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
    ring = None
    if through_ring:
        ring = simq_mod.DeviceReplayBuffer(64, cin)
        for t in zip(batch.state, batch.action, batch.reward, batch.next_state):
            ring.push(*t)

    def draw():
        if ring is None:
            return batch
        b = ring.gather(list(range(B)))
        assert b.ready_event is not None
        return b
    policy = simq_mod.FCN(cin, cout)
    target = simq_mod.FCN(cin, cout)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, wseed)))
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, wseed + 1000)))
    policy.train()
    target.eval()
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    from simq import _lib
    snap = simq_mod.FCN(cin, cout)           # (receives the post-step-1 state; built here: its initialisation copies from pageable memory and would stall the stream between the calls)
    torch.cuda.synchronize()
    _lib.lib.call('simq_launch_counts_reset')
    info1 = simq_mod.train(cfg, policy, target, opt, draw(), olearner.apply_transform, cases.GAMMA)
    ran = _lib.launch_counts()
    # what the checks of the first call read, cloned in stream order (no host synchronisation: the second call is enqueued while the first
    # one's backward pass still runs, as in bench.py's loop) ...
    last1 = {k: policy._last[k].clone() for k in ('q_sa', 'y')}
    tn1, grads1 = policy._simq_opt_state.total_norm.clone(), policy.flat_grads.clone()
    snap.flat_params.copy_(policy.flat_params)
    snap.bn_buffers.copy_(policy.bn_buffers)
    snap.num_batches_tracked = type(policy.num_batches_tracked)(policy.num_batches_tracked)
    # ... and the second call right behind it
    info2 = simq_mod.train(cfg, policy, target, opt, draw(), olearner.apply_transform, cases.GAMMA)
    last2 = {k: policy._last[k].clone() for k in ('q_sa', 'y')}
    assert ('_qtgt_bufs' in policy.__dict__) == through_ring          # the early stream (two alternating Q-map buffers) was taken only through the ring
    # the kernels bench.py times at this size are the ones compared here: Winograd planes through the exact-fp32 batched GEMM,
    # the image-tile 1x1 / strided convolutions, the fp32 stem; nothing of the bf16 families
    missing = [f for f in _FP32_FAMILIES_B32 if ran.get(f, 0) == 0]
    assert not missing and not [f for f in ran if 'bf16' in f or f.endswith('16')], (missing, ran)
    # the transform-domain contractions in the form the plan names (simq_plan_options.gemm_split) -- and in that form only
    gemm = {1: 'gemm_split3_batched', 0: 'gemm_f32_batched'}
    assert ran.get(gemm[policy.plan.options['gemm_split']], 0) > 0 and ran.get(gemm[1 - policy.plan.options['gemm_split']], 0) == 0, ran
    rel1 = lambda a, b: abs(a - b) / abs(b)
    assert rel1(info1['loss'], float(g['loss'][0])) < 1e-4 and rel1(info1['td_error'], float(g['td_error'][0])) < 1e-4
    q_sa, y = last1['q_sa'].cpu().double().numpy(), last1['y'].cpu().double().numpy()
    assert np.abs(q_sa - g['q_sa']).max() <= 1e-4 * np.abs(g['q_sa']).max()
    assert np.abs(y - g['y']).max() <= 1e-4 * np.abs(g['y']).max()
    # gradient (clipped in place) against the fp64 summary: total norm, per-tensor norms, sampled elements
    tn = float(tn1.item())
    assert rel1(tn, float(g['total_norm64'])) < 5e-2
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    gflat = grads1.detach().cpu().double() / coef
    keys = [str(k) for k in g['grad_keys']]
    got = {}
    for (pname, _, kind), (off, n, shape) in zip(policy._param_names, policy._grad_views):
        t = gflat[off:off + n].view(shape)
        got[arch.PREFIX + pname] = (t.permute(0, 3, 1, 2).contiguous() if len(shape) == 4 else t).reshape(-1)
    num = den = 0.0
    for i, k in enumerate(keys):
        flat = got[k]
        idx = torch.tensor(cases.sample_indices(flat.numel()))
        mine = np.concatenate([[float(flat.norm())], flat[idx].numpy()])
        num += ((mine[1:] - g['grad64'][i][1:]) ** 2).sum()
        den += (g['grad64'][i][1:] ** 2).sum()
        if g['grad64'][i][0] > 1e-3 * float(g['total_norm64']):          # tensors that carry gradient mass
            assert abs(mine[0] - g['grad64'][i][0]) <= 5e-2 * g['grad64'][i][0], k
    assert (num / den) ** 0.5 <= 5e-2, 'sampled-gradient rel-L2 error %.3g (reference fp32 itself: %.3g)' % ((num / den) ** 0.5, float(g['ref_fp32_grad_relerr']))
    sd1, sd_target = step2_oracle.snapshot(snap), step2_oracle.snapshot(target)
    # the second call per transition at 1e-4 against the fp64 oracle run from the HIP path's own post-step-1 state
    # (tests/step2_oracle.py); against the golden trajectory -- chaotic on these synthetic problems -- only a sanity bound
    step2_oracle.second_step_against_the_oracle(sd1, sd_target, batch, last2['q_sa'].cpu().numpy(), last2['y'].cpu().numpy(), info2)
    assert rel1(info2['loss'], float(g['loss'][1])) < 0.1 and rel1(info2['td_error'], float(g['td_error'][1])) < 0.1
    sd = policy.state_dict()
    assert all(int(sd[k]) == 4 for k in sd if k.endswith('num_batches_tracked'))
    spec = ofcn.state_spec(cin, cout)
    rows = np.asarray([[float(sd[k].double().sum()), float(sd[k].double().norm())] for k, _, kind in spec if ofcn.is_parameter(kind)])
    assert np.abs(rows[:, 1] - g['param_summary_after2'][:, 1]).max() <= 1e-4 * g['param_summary_after2'][:, 1].max()


@pytest.mark.parametrize('intention', [False, True], ids=['dqn', 'intention'])
def test_reference_style_main_loop_on_synthetic_env(simq_mod, tmp_path, intention):
    """train.py:main (:181-346) on the drop-ins with a synthetic environment (tools/train_synthetic.py): step, tracker,
    aliased device replay, train / train_intention, target sync, the train.py:294 debug path, checkpoints and resume."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('train_synthetic', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'tools', 'train_synthetic.py'))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    cfg = ts.default_cfg(use_predicted_intention=intention, num_input_channels=5 if intention else 4,
                         robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}] if intention else [{'lifting_robot': 2}],
                         total_timesteps=36)
    policy, log, policy_path, checkpoint_path = ts.run(cfg, str(tmp_path), verbose=False)
    assert len(log) >= 8
    keys = {'td_error', 'loss'} | ({'loss_intention'} if intention else set())
    assert all(set(info) == keys and all(np.isfinite(v) for v in info.values()) for _, _, info in log)
    # resume: the policy checkpoint loads into fresh nets (policies.py:25-33), the optimizer state into fresh optimizers
    cfg2 = ts.default_cfg(**{**vars(cfg), 'checkpoint_path': checkpoint_path, 'policy_path': policy_path, 'total_timesteps': 36})
    Policy = simq_mod.DQNIntentionPolicy if intention else simq_mod.DQNPolicy
    resumed = Policy(cfg2, train=True)
    for a, b in zip(policy.policy_nets, resumed.policy_nets):
        assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.bn_buffers, b.bn_buffers)
        assert a.num_batches_tracked == b.num_batches_tracked and b.training
    if intention:
        for a, b in zip(policy.intention_nets, resumed.intention_nets):
            assert torch.equal(a.flat_params, b.flat_params)
    ck = simq_mod.load_checkpoint(checkpoint_path)       # (the rings are pickled under the reference scripts' `__main__.*` names)
    opt = torch.optim.SGD(resumed.policy_nets[0].parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.load_state_dict(ck['optimizers'][0])
    assert len(opt.state) > 0
    # ... and the trainer itself continues from the files (train.py:197-209): rings back in HBM, timestep carried on
    assert len(ck['replay_buffers']) == len(cfg.robot_config) and all(len(b) > 8 for b in ck['replay_buffers'])
    cfg3 = ts.default_cfg(**{**vars(cfg2), 'total_timesteps': 44})
    _, log3, _, checkpoint3 = ts.run(cfg3, str(tmp_path), verbose=False)
    assert log3 and min(t for t, _, _ in log3) > 45 and all(np.isfinite(v) for _, _, info in log3 for v in info.values())
    assert os.path.basename(checkpoint3) == 'checkpoint_00000055.pth.tar' and not os.path.exists(checkpoint_path)


@pytest.mark.slow
def test_multiprocess_collector_round_robin_and_batched(simq_mod):
    """train_multiprocess.py:147-275 on the drop-ins: environments step in spawned worker processes (CPU only), the learner
    process serves them.  (1) Collector.step is the reference's round-robin and, for the same seeds, produces exactly the
    transitions the in-process collector produces for worker 0's environment; (2) Collector.step_all serves all workers from ONE
    batched forward (DQNPolicy.step_many), whose greedy actions equal per-environment policy.step; the transitions go into a
    device replay ring and a training step runs on them."""
    import types
    from simq.collector import Collector
    from simq.synth import synthetic_env_from_cfg
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}], num_input_channels=4, final_exploration=0.01,
                                checkpoint_path=None, policy_path=None, seed=3, episode_len=6, batch_size=8, use_double_dqn=True,
                                grad_norm_clipping=100)
    policy = simq_mod.DQNPolicy(cfg, train=True, random_seed=7)
    # step_many == step, environment by environment (greedy: eps = 0)
    envs = [synthetic_env_from_cfg(cfg, w) for w in range(3)]
    states = [e.reset() for e in envs]
    states[1][0][1] = None                                      # a robot that is not awaiting an action
    many = policy.step_many(states, exploration_eps=0.0)
    assert many == [policy.step(st, exploration_eps=0.0) for st in states]
    random.seed(5); a = policy.step_many(states, exploration_eps=0.5)
    random.seed(5); b = [policy.step(st, exploration_eps=0.5) for st in states]
    assert a == b                                               # same epsilon-greedy draw order as sequential step() calls

    # (1) round-robin over 2 worker processes vs the in-process collector (same env seeds for worker 0)
    col = Collector(cfg, policy, num_workers=2, env_fn=synthetic_env_from_cfg)
    ref = Collector(cfg, policy, num_workers=None, env_fn=synthetic_env_from_cfg)
    try:
        got0 = []
        for k in range(16):
            tr, done = col.step(0.0)
            if k % 2 == 0 and k >= 2:                           # call k serves worker k % 2 and returns the result of ITS previous
                got0.append((tr, done))                         # action, i.e. worker 0's env step k // 2 - 1
        want0 = [ref.step(0.0) for _ in range(7)]
        assert len(got0) == 7
        for (tg, dg), (tw, dw) in zip(got0, want0):
            assert dg == dw and len(tg) == len(tw)
            for bg, bw in zip(tg, tw):
                assert len(bg) == len(bw)
                for (s, a_, r, ns), (s2, a2, r2, ns2) in zip(bg, bw):
                    assert np.array_equal(s, s2) and a_ == a2 and r == r2 and ((ns is None) == (ns2 is None))
    finally:
        col.close()
        ref.close()

    # (2) batched service of 3 workers feeding device rings, then one training step per robot group
    col = Collector(cfg, policy, num_workers=3, env_fn=synthetic_env_from_cfg)
    rings = [simq_mod.DeviceReplayBuffer(256, 4) for _ in range(2)]
    try:
        for _ in range(14):
            for tr, done in col.step_all(0.3):
                for i, per_buffer in enumerate(tr):
                    for t in per_buffer:
                        rings[i].push(*t)
    finally:
        col.close()
    assert all(len(r) >= cfg.batch_size for r in rings)
    targets = policy.build_policy_nets()
    for i in range(2):
        targets[i].load_state_dict(policy.policy_nets[i].state_dict()); targets[i].eval()
        opt = torch.optim.SGD(policy.policy_nets[i].parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        info = simq_mod.train(cfg, policy.policy_nets[i], targets[i], opt, rings[i].sample(cfg.batch_size), policy.apply_transform, 0.85)
        assert np.isfinite(info['loss']) and np.isfinite(info['td_error'])


@pytest.mark.slow
def test_multiprocess_trainer_loop_on_synthetic_envs(simq_mod, tmp_path):
    """train_multiprocess.py:main on the drop-ins (tools/train_synthetic.py::run_multiprocess): 3 spawned environment processes,
    batched service, HBM rings, training, target sync, checkpoints -- and a resumed second run continues from the files."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('train_synthetic', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'tools', 'train_synthetic.py'))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    cfg = ts.default_cfg(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}], total_timesteps=36, episode_len=8)
    policy, log, policy_path, checkpoint_path = ts.run_multiprocess(cfg, str(tmp_path), num_workers=3, verbose=False)
    assert len(log) >= 8 and all(np.isfinite(v) for _, _, info in log for v in info.values())
    assert {i for _, i, _ in log} == {0, 1}                         # both robot groups trained
    ck = simq_mod.load_checkpoint(checkpoint_path)
    assert ck['timestep'] >= 45 and len(ck['replay_buffers']) == 2 and all(len(b) > 8 for b in ck['replay_buffers'])
    cfg2 = ts.default_cfg(**{**vars(cfg), 'checkpoint_path': checkpoint_path, 'policy_path': policy_path, 'total_timesteps': 44})
    _, log2, _, checkpoint2 = ts.run_multiprocess(cfg2, str(tmp_path), num_workers=2, verbose=False)
    assert log2 and min(t for t, _, _ in log2) > ck['timestep'] and os.path.exists(checkpoint2) and not os.path.exists(checkpoint_path)
