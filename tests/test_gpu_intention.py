"""GPU: the intention-prediction path (DQNIntentionPolicy, policies.py:76-146; train_intention, train.py:143-158)
through libsimq against the golden fixtures (reference-pinned) and the CPU oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import cases
from oracle import fcn as ofcn
from oracle import learner as olearner
from simq import arch, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def simq_mod():
    import simq
    from simq import _lib  # noqa: F401
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return simq


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_bce_split_concat_kernels(simq_mod):
    from simq._lib import lib, ptr, stream_ptr
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    for n in (1, 255, 96 * 96 * 3 + 5):
        x = (torch.randn(n, generator=g) * 6).float()
        x[0] = 0.0
        if n > 2:
            x[1], x[2] = 60.0, -60.0                       # saturated logits: the stable form must not overflow
        t = torch.rand(n, generator=g)
        xd, td = x.to(dev), t.to(dev)
        dx = torch.empty_like(xd)
        ls = torch.zeros(1, dtype=torch.float64, device=dev)
        lib.call('simq_bce_with_logits', ptr(xd), ptr(td), n, ptr(dx), ptr(ls), stream_ptr(dev))
        x64 = x.double().requires_grad_(True)
        ref = torch.nn.functional.binary_cross_entropy_with_logits(x64, t.double())
        ref.backward()
        assert abs(float(ls.item()) / n - float(ref.detach())) <= 1e-6 * abs(float(ref.detach()))
        assert rel(dx, x64.grad) < 1e-5
        lib.call('simq_bce_with_logits', ptr(xd), ptr(td), n, None, ptr(ls), stream_ptr(dev))     # loss only
        assert abs(float(ls.item()) / n - float(ref.detach())) <= 1e-6 * abs(float(ref.detach()))
    s = torch.rand(7, 96, 96, 5, generator=g)
    sd = s.to(dev)
    head = torch.empty(7, 96, 96, 4, device=dev)
    last = torch.empty(7, 96, 96, device=dev)
    lib.call('simq_split_last_channel', ptr(sd), ptr(head), ptr(last), 7 * 96 * 96, 5, stream_ptr(dev))
    assert torch.equal(head.cpu(), s[..., :4]) and torch.equal(last.cpu(), s[..., 4])
    logit = torch.randn(96, 96, generator=g).to(dev)
    out = torch.empty(1, 96, 96, 5, device=dev)
    prob = torch.empty(96, 96, device=dev)
    lib.call('simq_sigmoid_concat', ptr(head[:1].contiguous()), ptr(logit), ptr(out), ptr(prob), 96 * 96, 4, stream_ptr(dev))
    assert torch.equal(out[0, ..., :4].cpu(), s[0, ..., :4])
    assert rel(out[0, ..., 4], torch.sigmoid(logit.double())) < 1e-6 and torch.equal(out[0, ..., 4], prob)
    with pytest.raises(Exception):
        lib.call('simq_split_last_channel', ptr(sd), ptr(head), ptr(last), 7 * 96 * 96, 1, stream_ptr(dev))


@pytest.mark.parametrize('case', cases.INTENTION_CASES, ids=[c[0] for c in cases.INTENTION_CASES])
def test_train_intention_vs_golden_and_oracle(simq_mod, case, golden_dir):
    name, cin_full, B, wseed, dseed = case
    cin = cin_full - 1
    g = np.load('%s/%s.npz' % (golden_dir, name))
    batch = cases.make_batch(cin_full, 1, B, dseed)
    spec = ofcn.state_spec(cin, 1)
    net = simq_mod.FCN(cin, 1)
    net.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, 1, wseed)))
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    st64 = cases.oracle_state(cin, 1, wseed, torch.float64)
    ex64 = {}
    olearner.train_intention_step(st64, spec, [None] * len(olearner.grad_keys(spec)), batch, cases.LR, cases.MOMENTUM,
                                  cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
    info1 = simq_mod.train_intention(net, opt, batch, olearner.apply_transform)
    assert set(info1) == {'loss_intention'} and isinstance(info1['loss_intention'], float)
    assert rel(info1['loss_intention'], g['loss_intention'][0]) < TOL
    assert rel(net._last['logits'], g['output_step1']) < TOL
    # gradient (unclipped on this path) vs the fp64 oracle; same ill-conditioning bar as the TD step
    got = {}
    gflat = net.flat_grads.detach().cpu()
    for (pname, _, kind), (off, n, shape) in zip(net._param_names, net._grad_views):
        t = gflat[off:off + n].view(shape)
        got[arch.PREFIX + pname] = t.permute(0, 3, 1, 2).contiguous() if len(shape) == 4 else t
    num = sum(float((got[k].double() - ex64['grads'][k]).pow(2).sum()) for k in ex64['grads'])
    den = sum(float(ex64['grads'][k].pow(2).sum()) for k in ex64['grads'])
    err = (num / den) ** 0.5
    assert err <= 5e-2, 'gradient rel-L2 error %.3g vs fp64 (reference fp32 itself: %.3g)' % (err, float(g['ref_fp32_grad_relerr']))
    p0 = next(iter(net.parameters()))
    assert opt.state[p0]['momentum_buffer'].data_ptr() == net._simq_opt_state.momentum.data_ptr()
    info2 = simq_mod.train_intention(net, opt, batch, olearner.apply_transform)
    assert rel(info2['loss_intention'], g['loss_intention'][1]) < 2e-2
    sd = net.state_dict()
    assert all(int(sd[k]) == 2 for k in sd if k.endswith('num_batches_tracked'))
    rows = np.asarray([[float(sd[k].double().sum()), float(sd[k].double().norm())] for k, _, kind in spec if ofcn.is_parameter(kind)])
    assert np.abs(rows[:, 1] - g['param_summary_after2'][:, 1]).max() <= 1e-4 * g['param_summary_after2'][:, 1].max()
    # device-resident batch (DeviceReplayBuffer.sample) takes the same path
    buf = simq_mod.DeviceReplayBuffer(8, cin_full)
    for s, a, r, ns in zip(batch.state, batch.action, batch.reward, batch.next_state):
        buf.push(s, a, r, ns)
    net2 = simq_mod.FCN(cin, 1)
    net2.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, 1, wseed)))
    net2.train()
    from simq.learner import train_intention_step
    got2 = train_intention_step(net2, buf.gather(list(range(B))), cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY)
    assert rel(got2['loss_intention'], g['loss_intention'][0]) < TOL
    with pytest.raises(Exception):
        simq_mod.train_intention(simq_mod.FCN(cin, 2), opt, batch, None)          # not an intention head
    with pytest.raises(Exception):
        simq_mod.train_intention(simq_mod.FCN(cin + 1, 1), opt, batch, None)      # channel mismatch


def test_intention_policy_step_golden(simq_mod, golden_dir):
    g = np.load('%s/intention_step.npz' % golden_dir)
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 1}, {'pushing_robot': 1}], num_input_channels=5,
                                final_exploration=0.01, checkpoint_path=None)
    pol = simq_mod.DQNIntentionPolicy(cfg, train=False, random_seed=9)
    assert [n.num_input_channels for n in pol.intention_nets] == [4, 4]
    assert [n.num_output_channels for n in pol.intention_nets] == [1, 1]
    # the oracle fixture built its nets in the order policy[0], policy[1], intention[0], intention[1] (seeds 71..74)
    for i, (net, co) in enumerate(zip(pol.policy_nets + pol.intention_nets, (2, 1, 1, 1))):
        net.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(net.num_input_channels, co, 71 + i)))
    s = synth.make_states(2, 4, 81)
    a, info = pol.step([[s[0]], [s[1]]], exploration_eps=0.0, debug=True)
    assert rel(np.stack([info['output_intention'][0][0], info['output_intention'][1][0]]), g['output_intention']) < TOL
    assert rel(np.stack([info['state_intention'][0][0], info['state_intention'][1][0]]), g['state_intention']) < TOL
    assert rel(info['output'][0][0], g['q0']) < TOL and rel(info['output'][1][0], g['q1']) < TOL
    assert [a[0][0], a[1][0]] == g['actions'].tolist()
    # non-debug path keeps the predicted map in HBM between the two nets: same actions
    assert pol.step([[s[0]], [s[1]]], exploration_eps=0.0) == a
    # epsilon-greedy draws: the actions the reference's own DQNIntentionPolicy returned under random.seed(123)
    import random as _random
    _random.seed(123)
    o1 = [pol.step([[s[0]], [s[1]]], exploration_eps=0.5) for _ in range(3)]
    assert [[x[0][0], x[1][0]] for x in o1] == g['eps_half_actions'].tolist()
    si = pol.step_intention([[s[0]], [None]])
    assert si[1][0] is None and si[0][0].shape == (96, 96, 5) and np.array_equal(si[0][0][:, :, :4], s[0])
    # train mode: ground-truth channel dropped (predicted) or used (policies.py:120-131); nets return to train mode
    pol.train = True
    for n in pol.policy_nets + pol.intention_nets:
        n.train()
    full = np.concatenate([s[0], np.zeros((96, 96, 1), np.float32)], axis=2)
    assert pol.step([[full], [None]], exploration_eps=0.0)[0][0] == a[0][0]
    a_gt = pol.step([[full], [None]], exploration_eps=0.0, use_ground_truth_intention=True)
    assert 0 <= a_gt[0][0] < 2 * 96 * 96
    assert all(n.training for n in pol.policy_nets + pol.intention_nets)
    # step_many on the predicted-intention path: one batched intention forward + one batched Q forward per robot group for all
    # environments == environment-by-environment step() (greedy), and the same epsilon-greedy draws under a seed
    s3 = synth.make_states(5, 4, 83)
    fulls = [np.concatenate([x, np.zeros((96, 96, 1), np.float32)], axis=2) for x in s3]
    envs = [[[fulls[0]], [fulls[1]]], [[fulls[2]], [None]], [[None], [fulls[3]]], [[fulls[4]], [fulls[0]]]]
    many = pol.step_many(envs, exploration_eps=0.0)
    assert many == [pol.step(st, exploration_eps=0.0) for st in envs]
    import random
    random.seed(3); m1 = pol.step_many(envs, exploration_eps=0.5)
    random.seed(3); m2 = [pol.step(st, exploration_eps=0.5) for st in envs]
    assert m1 == m2
    assert all(n.training for n in pol.policy_nets + pol.intention_nets)
    pol.train = False
    for n in pol.policy_nets + pol.intention_nets:
        n.eval()
    envs4 = [[[s3[0]], [s3[1]]], [[s3[2]], [s3[3]]]]
    assert pol.step_many(envs4, exploration_eps=0.0) == [pol.step(st, exploration_eps=0.0) for st in envs4]


def test_intention_checkpoint_roundtrip(simq_mod, tmp_path):
    """train.py:320-321 / policies.py:80-83: 'state_dicts_intention' in the policy checkpoint."""
    cfg = types.SimpleNamespace(robot_config=[{'pushing_robot': 2}], num_input_channels=5, final_exploration=0.01,
                                checkpoint_path=None)
    pol = simq_mod.DQNIntentionPolicy(cfg, train=True, random_seed=1)
    path = tmp_path / 'policy_00000001.pth.tar'
    torch.save({'timestep': 1, 'state_dicts': [n.state_dict() for n in pol.policy_nets],
                'state_dicts_intention': [n.state_dict() for n in pol.intention_nets]}, str(path))
    cfg2 = types.SimpleNamespace(**{**vars(cfg), 'checkpoint_path': 'x', 'policy_path': str(path)})
    pol2 = simq_mod.DQNIntentionPolicy(cfg2, train=False)
    assert torch.equal(pol2.intention_nets[0].flat_params, pol.intention_nets[0].flat_params)
    assert torch.equal(pol2.policy_nets[0].flat_params, pol.policy_nets[0].flat_params)
    assert not pol2.intention_nets[0].training
