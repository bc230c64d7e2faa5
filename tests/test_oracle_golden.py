"""CPU: the oracle restatement reproduces the committed golden fixtures.

The fixtures were written by oracle/gen_golden.py AFTER a bit-exact comparison
with the imported reference (build container).  Here (any host, no reference) the
oracle is re-run and must agree within fp32 round-off -- a different CPU may pick
different MKL-DNN kernels, so the check is 1e-5 (max-abs / max-abs), not bitwise.
"""
import random

import numpy as np
import pytest
import torch

from oracle import cases, fcn, learner
from oracle import policy as opolicy
from simq import arch, synth


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_state_spec_matches_product_table():
    for cin, cout in ((4, 2), (5, 1), (10, 2)):
        assert fcn.state_spec(cin, cout) == arch.state_spec(cin, cout)
        assert len(fcn.state_spec(cin, cout)) == 138
        n_grad = sum(int(np.prod(s)) for _, s, k in arch.state_spec(cin, cout) if k in arch.TRAINABLE_KINDS)
        assert n_grad == {(4, 2): 11249826, (5, 1): 11252929, (10, 2): 11249826 + 6 * 64 * 49}[(cin, cout)]


@pytest.mark.parametrize('case', cases.FORWARD_CASES, ids=[c[0] for c in cases.FORWARD_CASES])
def test_forward_golden(case, golden_dir):
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    x = torch.cat([learner.apply_transform(s) for s in synth.make_states(B, cin, dseed)])
    st = cases.oracle_state(cin, cout, wseed)
    taps = {}
    with torch.no_grad():
        q = fcn.fcn_forward(st, x, False, taps)
    assert rel(q.numpy(), g['q_eval']) < 1e-5
    for k, t in taps.items():
        ref = g['tap_eval.' + k]
        assert abs(float(t.double().abs().mean()) - ref[1]) <= 1e-5 * ref[1]
    st = cases.oracle_state(cin, cout, wseed)
    with torch.no_grad():
        q = fcn.fcn_forward(st, x, True)
    assert rel(q.numpy(), g['q_train']) < 1e-5
    assert rel(cases.bn_buffer_vector(st), g['bn_buffers_after']) < 1e-5
    assert all(int(st[k]) == 1 for k in st if k.endswith('num_batches_tracked'))


@pytest.mark.parametrize('case', cases.TRAIN_CASES[:2] + cases.TRAIN_CASES_CIN, ids=[c[0] for c in cases.TRAIN_CASES[:2] + cases.TRAIN_CASES_CIN])
def test_train_step_golden(case, golden_dir):
    name, cin, cout, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    cfg, batch, spec = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed), fcn.state_spec(cin, cout)
    st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
    mom = [None] * len(learner.grad_keys(spec))
    ex = [{}, {}]
    info = [learner.train_step(cfg, st, tg, spec, mom, batch, cases.GAMMA, cases.LR, cases.MOMENTUM,
                               cases.WEIGHT_DECAY, extras=ex[i]) for i in range(2)]
    assert rel([i['loss'] for i in info], g['loss']) < 2e-5
    assert rel([i['td_error'] for i in info], g['td_error']) < 2e-5
    assert rel(ex[0]['q'].numpy(), g['q_sa']) < 2e-5
    assert rel(ex[0]['y'].numpy(), g['y']) < 2e-5
    # one train() call bumps num_batches_tracked by 2 (train.py:114 + :121) -> 4 after two calls
    assert (g['num_batches_tracked'] == 4).all()
    assert all(int(st[k]) == 4 for k in st if k.endswith('num_batches_tracked'))
    assert all(int(tg[k]) == 0 for k in tg if k.endswith('num_batches_tracked'))
    assert rel(cases.bn_buffer_vector(st), g['bn_buffers_after2']) < 1e-4
    # gradient: judged against the fp64 golden relative to the reference's own fp32 error
    g32 = cases.grad_summary(ex[0]['grads'])
    num = sum(((g32[k][1:] - g['grad64'][i][1:]) ** 2).sum() for i, k in enumerate(g32))
    den = sum((g['grad64'][i][1:] ** 2).sum() for i, k in enumerate(g32))
    ref_err = float(g['ref_fp32_grad_relerr'])
    assert (num / den) ** 0.5 < max(3 * ref_err, 1e-3)
    # fc.* never receive a gradient; head conv1/conv2 biases have a ~0 gradient (BN follows)
    assert not any(k.endswith('fc.weight') or k.endswith('fc.bias') for k in g32)
    assert g32['module.conv1.bias'][0] < 1e-5 and g32['module.conv2.bias'][0] < 1e-5


def test_sampler_golden(golden_dir):
    g = np.load('%s/sampler.npz' % golden_dir)
    for n, B, seed in cases.SAMPLER_CASES:
        buf = learner.ReplayBuffer(n)
        for i in range(n + 3):
            buf.push(i, i, float(i), None)
        assert len(buf) == n and buf.position == 3 % n
        random.seed(seed)
        picked = buf.sample(B)
        assert list(picked.state) == list(g['n%d_b%d_s%d' % (n, B, seed)])
        random.seed(seed)
        idx = random.sample(range(n), B)
        assert idx == list(g['idx_n%d_b%d_s%d' % (n, B, seed)])
        assert [buf.buffer[i].state for i in idx] == list(picked.state)


def test_policy_step_golden(golden_dir):
    import types
    g = np.load('%s/policy_step.npz' % golden_dir)
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}], num_input_channels=4,
                                final_exploration=0.01)
    seeds = iter([51, 52])
    pol = opolicy.DQNPolicy(cfg, lambda ci, co: cases.oracle_state(ci, co, next(seeds)), train=False, random_seed=5)
    s = synth.make_states(3, 4, 61)
    state = [[s[0], None], [s[1]]]
    acts = [pol.step(state, exploration_eps=e) for e in (0.0, 0.5, 1.0, 0.5)]
    assert [[a[0][0], a[1][0]] for a in acts] == g['actions'].tolist()
    assert all(a[0][1] is None for a in acts)
    a, info = pol.step([[None, s[2]], [None]], exploration_eps=0.0, debug=True)
    assert a[0][1] == int(g['debug_action'][0]) and a[1][0] is None
    assert rel(info['output'][0][1], g['debug_output']) < 1e-5
    assert int(np.argmax(g['debug_output'].reshape(-1))) == a[0][1]      # first-index argmax, CHW order


def test_argmax_first_index_tiebreak():
    q = torch.zeros(3, 2 * 96 * 96)
    q[0, 5] = q[0, 9000] = 1.0
    q[1, 18431] = 2.0
    assert q.max(1)[1].tolist() == [5, 18431, 0]


@pytest.mark.parametrize('case', cases.INTENTION_CASES, ids=[c[0] for c in cases.INTENTION_CASES])
def test_train_intention_golden(case, golden_dir):
    """train.train_intention (train.py:143-158): fixture written after a bit-exact match with the reference."""
    name, cin_full, B, wseed, dseed = case
    g = np.load('%s/%s.npz' % (golden_dir, name))
    batch, spec = cases.make_batch(cin_full, 1, B, dseed), fcn.state_spec(cin_full - 1, 1)
    st = cases.oracle_state(cin_full - 1, 1, wseed)
    mom = [None] * len(learner.grad_keys(spec))
    ex = [{}, {}]
    info = [learner.train_intention_step(st, spec, mom, batch, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, extras=ex[i])
            for i in range(2)]
    assert all(set(i) == {'loss_intention'} for i in info)
    # bit-exact on the host that wrote the fixture; other CPUs select different MKL-DNN kernels (fp32 summation order), and the
    # second call's loss sees the first update: bars as for the other train-mode fixtures
    assert rel([i['loss_intention'] for i in info], g['loss_intention']) < 5e-4
    assert rel(ex[0]['output'].numpy(), g['output_step1']) < 2e-4
    assert all(int(st[k]) == 2 for k in st if k.endswith('num_batches_tracked'))      # one train-mode forward per call
    g32 = cases.grad_summary(ex[0]['grads'])
    num = sum(((g32[k][1:] - g['grad64'][i][1:]) ** 2).sum() for i, k in enumerate(g32))
    den = sum((g['grad64'][i][1:] ** 2).sum() for i, k in enumerate(g32))
    assert (num / den) ** 0.5 < max(3 * float(g['ref_fp32_grad_relerr']), 1e-3)


def test_intention_policy_step_golden(golden_dir):
    import types
    g = np.load('%s/intention_step.npz' % golden_dir)
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 1}, {'pushing_robot': 1}], num_input_channels=5,
                                final_exploration=0.01)
    seeds = iter([71, 72, 73, 74])
    pol = opolicy.DQNIntentionPolicy(cfg, lambda ci, co: cases.oracle_state(ci, co, next(seeds)), train=False, random_seed=9)
    s = synth.make_states(2, 4, 81)
    a, info = pol.step([[s[0]], [s[1]]], exploration_eps=0.0, debug=True)
    assert [a[0][0], a[1][0]] == g['actions'].tolist()
    assert rel(np.stack([info['output_intention'][0][0], info['output_intention'][1][0]]), g['output_intention']) < 1e-4   # host-independent bar
    si = info['state_intention'][0][0]
    assert si.shape == (96, 96, 5) and np.array_equal(si[:, :, :4], s[0]) and 0.0 <= si[:, :, 4].min() <= si[:, :, 4].max() <= 1.0
    # epsilon-greedy draws of the reference's own class under random.seed(123) (policies.py:61-62 through :119-146)
    import random
    random.seed(123)
    o1 = [pol.step([[s[0]], [s[1]]], exploration_eps=0.5) for _ in range(3)]
    assert [[x[0][0], x[1][0]] for x in o1] == g['eps_half_actions'].tolist()
    # train-mode policy: the ground-truth map (last channel) is dropped before predicting, or used as is
    pol.train = True
    full = np.concatenate([s[0], np.zeros((96, 96, 1), np.float32)], axis=2)
    a_pred = pol.step([[full], [None]], exploration_eps=0.0)
    assert a_pred[0][0] == a[0][0]
    a_gt = pol.step([[full], [None]], exploration_eps=0.0, use_ground_truth_intention=True)
    assert 0 <= a_gt[0][0] < 2 * 96 * 96


def test_full_size_golden_is_present_and_consistent(golden_dir):
    """tests/golden/train_c4o2_b32.npz (BASELINE configs[1] size; the oracle needs ~40 s there, so it is not re-run on
    the CPU tier): internal consistency of the stored summaries."""
    name, cin, cout, B, wseed, dseed = cases.TRAIN_CASES_FULL[0]
    g = np.load('%s/%s.npz' % (golden_dir, name))
    assert g['q_sa'].shape == (B,) and g['y'].shape == (B,) and g['loss'].shape == (2,)
    d = g['q_sa'].astype(np.float64) - g['y'].astype(np.float64)
    huber = np.where(np.abs(d) < 1, 0.5 * d * d, np.abs(d) - 0.5).mean()
    assert abs(huber - g['loss'][0]) <= 1e-5 * abs(g['loss'][0])                    # train.py:129 on the stored q, y
    assert abs(np.abs(d).mean() - g['td_error'][0]) <= 1e-5 * abs(g['td_error'][0])  # train.py:127,138
    assert abs(g['loss64'] - g['loss'][0]) <= 1e-4 * abs(g['loss64'])
    tot = np.sqrt((g['grad64'][:, 0] ** 2).sum())
    assert abs(tot - g['total_norm64']) <= 1e-9 * g['total_norm64']                  # global norm = norm of tensor norms
    assert float(g['ref_fp32_grad_relerr']) < 5e-3
    assert (g['num_batches_tracked'] == 4).all()


@pytest.mark.parametrize('name', ['dp_c5o2_b8_w8', 'dp_c5o1_b8_w2'])
def test_data_parallel_emulation_golden(name, golden_dir):
    """Fixture G7: written after oracle.learner.dp_emulation agreed BIT FOR BIT with the reference's own FCN run replica
    by replica (nn.DataParallel semantics, policies.py:39).  w8: eight 1-transition shards, three of them all-terminal."""
    _, cin, cout, gB, world, wseed, dseed = [c for c in cases.DP_CASES if c[0] == name][0]
    g = np.load('%s/%s.npz' % (golden_dir, name))
    cfg, batch, spec = cases.make_cfg(gB), cases.make_batch(cin, cout, gB, dseed), fcn.state_spec(cin, cout)
    st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
    total, loss, td = learner.dp_emulation(cfg, st, tg, spec, batch, world, cases.GAMMA)
    assert abs(loss - float(g['loss'])) <= 1e-5 * abs(float(g['loss'])) and abs(td - float(g['td_error'])) <= 1e-5 * abs(float(g['td_error']))
    assert rel(cases.bn_buffer_vector(st), g['bn_buffers_after']) < 1e-5
    # the gradient of these tiny shards is conditioned as DESIGN section 2 describes: hold the fp32 oracle to the fp64 summary
    # within a few times the error the reference's fp32 had on the generating host
    off, num, den = 0, 0.0, 0.0
    for i, k in enumerate(learner.grad_keys(spec)):
        n = st[k].numel()
        flat = total[off:off + n].double()
        off += n
        idx = torch.tensor(cases.sample_indices(n))
        num += float(((flat[idx].numpy() - g['grad64'][i][1:]) ** 2).sum())
        den += float((g['grad64'][i][1:] ** 2).sum())
    assert (num / den) ** 0.5 <= max(10 * float(g['ref_fp32_grad_relerr']), 5e-3)
    assert int(g['all_terminal_shards']) == (3 if name.endswith('w8') else 0)
    assert float(g['shard_vs_single_replica_relerr']) > 0.5      # per-replica BN statistics are NOT SyncBN


def test_bf16_points_model_is_the_fp64_oracle_when_its_points_are_off():
    """oracle/bf16_points.py (the rounded-operand model the GPU bf16 tests compare against): with every rounding point switched off it IS
    oracle.learner.train_step / oracle.fcn.fcn_forward in fp64 -- loss, TD targets, gradient, BatchNorm buffers at 1e-12 -- and with the points
    on it differs (the roundings are real) while staying inside the reference's own bf16-autocast calibration of these small batches
    (fixture G8: train-mode Q-maps 4.6-8.7e-2).  The committed fixtures tests/golden/bf16pts_*.npz carry the model's output."""
    import numpy as np
    from oracle import bf16_points as bp
    cin, cout, B, ws, ds = 4, 2, 3, 31, 41
    cfg, batch, spec = cases.make_cfg(B), cases.make_batch(cin, cout, B, ds), fcn.state_spec(cin, cout)
    gk = learner.grad_keys(spec)

    def run(fn, **kw):
        st, tg = cases.oracle_state(cin, cout, ws, torch.float64), cases.oracle_state(cin, cout, ws + 1000, torch.float64)
        ex = {}
        info = fn(cfg, st, tg, spec, [None] * len(gk), batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, extras=ex, **kw)
        return info, ex, st
    i0, e0, s0 = run(learner.train_step, dtype=torch.float64)
    i1, e1, s1 = run(bp.train_step, points=False)
    i2, e2, s2 = run(bp.train_step, points=True)
    cat = lambda e: torch.cat([g.reshape(-1) for g in e['grads'].values()])
    assert abs(i1['loss'] - i0['loss']) <= 1e-12 * abs(i0['loss']) and abs(i1['td_error'] - i0['td_error']) <= 1e-12 * abs(i0['td_error'])
    assert float((cat(e1) - cat(e0)).norm() / cat(e0).norm()) < 1e-10
    assert float((e1['output'] - e0['output']).abs().max() / e0['output'].abs().max()) < 1e-12
    assert np.abs(cases.bn_buffer_vector(s1) - cases.bn_buffer_vector(s0)).max() < 1e-12
    dq = float((e2['output'] - e0['output']).abs().max() / e0['output'].abs().max())
    assert 1e-4 < dq < 0.2, dq
    for k in s2:
        if k.endswith('num_batches_tracked'):
            assert int(s2[k]) == int(s0[k])
