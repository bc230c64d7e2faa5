"""GPU: the stream structure of one TD step (round 4) changes WHEN kernels run, never what they compute.

simq_train_step runs the three forwards of train.py:114-122 side by side (the policy's no-grad forward on a third stream with its
BatchNorm running-statistics update deferred and applied behind the grad-mode forward's, in the reference's order) and the weight
gradients of the residual blocks on a side stream up to one block behind the dgrads (a second set of gradient temporaries).  On a
DETERMINISTIC plan (fixed-order reductions everywhere) every result of two consecutive steps -- losses, gradient, parameters and the
BatchNorm buffers, whose update order is exactly what the deferral has to preserve -- must equal the fully serial order BIT FOR BIT,
for every setting of the two scheduling options of the plan (simq_plan_options.wgrad_overlap / fwd_overlap: round 5 -- they were
process-global switches before); the standalone backward entry points (FCN.backward, simq_backward*) take a side stream the PLAN owns
(created on first use, destroyed with the plan) and are held to the same bar.  Against the reference itself the overlapped step is what every golden test
of tests/test_gpu_fcn.py / test_gpu_sized.py runs (the defaults)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def env():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    import simq
    import simq.learner as sl
    from oracle import cases, fcn as ofcn
    from simq import synth
    from simq._lib import lib
    return dict(simq=simq, sl=sl, cases=cases, ofcn=ofcn, synth=synth, lib=lib)


def _nets(e, precision, options, cin=5, cout=2):
    policy = e['simq'].FCN(cin, cout, precision=precision, options=options)
    target = e['simq'].FCN(cin, cout, precision=precision, options=options)
    policy.load_state_dict(e['ofcn'].state_from_numpy(e['synth'].make_state_dict(cin, cout, 3)))
    target.load_state_dict(e['ofcn'].state_from_numpy(e['synth'].make_state_dict(cin, cout, 4)))
    policy.train(); target.eval()
    return policy, target


def _two_steps(e, wgrad, fwd, B, precision='fp32', options=None, cin=5, cout=2):
    c = e['cases']
    opts = dict(options if options is not None else {'deterministic': 1}, wgrad_overlap=wgrad, fwd_overlap=fwd)
    policy, target = _nets(e, precision, opts, cin, cout)
    assert policy.plan.options['wgrad_overlap'] == wgrad and policy.plan.options['fwd_overlap'] == fwd
    losses = []
    for s in range(2):     # (the second step re-records every event and reuses both sets of temporaries)
        info = e['sl'].train_step(policy, target, c.make_batch(cin, cout, B, 7 + s), c.GAMMA, B, c.LR, c.MOMENTUM, c.WEIGHT_DECAY, c.CLIP,
                                  use_double_dqn=True)
        losses.append((info['loss'], info['td_error']))
    torch.cuda.synchronize()
    return dict(loss=losses, grads=policy.flat_grads.clone(), params=policy.flat_params.clone(), bn=policy.bn_buffers.clone())


@pytest.mark.parametrize('B', [6, 32])
def test_overlapped_step_equals_the_serial_step_bit_for_bit(env, B):
    ref = _two_steps(env, 0, 0, B)                     # weight gradients behind the dgrads, the policy's forwards one after the other
    assert all(l == l and abs(l) < 1e6 for pair in ref['loss'] for l in pair)
    for wgrad, fwd in ((4, 2), (1, 2), (3, 1), (4, 0), (0, 2)):
        r = _two_steps(env, wgrad, fwd, B)
        assert r['loss'] == ref['loss'], (wgrad, fwd, r['loss'], ref['loss'])
        assert torch.equal(r['bn'], ref['bn']), 'BatchNorm buffers differ (wgrad_overlap %d, fwd_overlap %d): the deferred update is out of order' % (wgrad, fwd)
        assert torch.equal(r['grads'], ref['grads']), 'gradient differs (wgrad_overlap %d, fwd_overlap %d)' % (wgrad, fwd)
        assert torch.equal(r['params'], ref['params']), 'parameters differ (wgrad_overlap %d, fwd_overlap %d)' % (wgrad, fwd)


@pytest.mark.parametrize('precision,B', [('bf16', 16), ('bf16', 128), ('bf16x3', 16)])
def test_overlapped_bf16_step_keeps_the_buffers_of_the_serial_step(env, precision, B):
    # bf16: the three forwards side by side; the weight gradients up to one block behind the dgrads on planes of their own (4, the default)
    # or beside them until the end of the block (2)
    ref = _two_steps(env, 0, 0, B, precision=precision)
    for wgrad, fwd in ((4, 2), (2, 2), (4, 0)):
        r = _two_steps(env, wgrad, fwd, B, precision=precision)
        assert r['loss'] == ref['loss'], (wgrad, fwd)
        for k in ('bn', 'grads', 'params'):
            assert torch.equal(r[k], ref[k]), (k, wgrad, fwd)


@pytest.mark.parametrize('precision,B,modes', [('fp32', 8, (0, 4, 1)), ('bf16', 16, (0, 4, 2))])
def test_standalone_backward_on_the_plan_side_stream(env, precision, B, modes):
    # FCN.backward / simq_backward_phase: dense upstream gradient, weight gradients on the plan-owned side stream (wgrad_overlap = 0: none)
    from simq._lib import MODE_TRAIN
    e = env
    cin, cout = 4, 2
    out = {}
    for wgrad in modes:
        policy, _ = _nets(e, precision, {'deterministic': 1, 'wgrad_overlap': wgrad}, cin, cout)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 96, 96, cin, generator=g).cuda()
        for _ in range(2):
            q = policy._forward_raw(x, MODE_TRAIN)
            dq = (torch.randn(q.shape, generator=g) * 1e-3).cuda()
            policy._backward_raw(dq, B)
        torch.cuda.synchronize()
        out[wgrad] = (q.detach().clone(), policy.flat_grads.clone())
    assert float(out[0][1].abs().max()) > 0
    for wgrad in modes[1:]:
        assert torch.equal(out[wgrad][0], out[0][0])
        assert torch.equal(out[wgrad][1], out[0][1]), 'standalone backward: gradient differs with wgrad_overlap %d' % wgrad



def test_plan_owned_streams_are_created_lazily_and_die_with_the_plan(env):
    """Round 5: the third stream of fwd_overlap = 2, the side stream of a standalone backward and the loss-copy stream belong to the PLAN
    (one set per device), not to the calling host thread: creating and dropping many nets must not grow the process's stream count, and a
    backward called from ANOTHER host thread (torch's autograd engine does that) uses the same plan-owned stream."""
    import gc
    import threading
    from simq._lib import MODE_TRAIN
    e = env
    B, cin, cout = 4, 4, 2
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 96, 96, cin, generator=g).cuda()
    ref = None
    for rep in range(6):                            # (6 plans, each creating its side stream and 4 events; a leak shows in rocm-smi, a crash here)
        policy, _ = _nets(e, 'fp32', {'deterministic': 1}, cin, cout)
        q = policy._forward_raw(x, MODE_TRAIN)
        dq = torch.full_like(q, 1e-3)
        err = []

        def run():
            try:
                torch.cuda.set_device(0)
                policy._backward_raw(dq, B)
                torch.cuda.synchronize()
            except Exception as ex:               # noqa: BLE001
                err.append(ex)
        t = threading.Thread(target=run)
        t.start(); t.join()
        assert not err, err
        grads = policy.flat_grads.clone()
        if ref is None:
            ref = grads
        assert torch.equal(grads, ref)
        del policy, _
        gc.collect()


@pytest.mark.parametrize('precision,B,after_block', [('fp32', 8, 4), ('bf16', 16, 4), ('fp32', 8, -1), ('bf16', 16, 0), ('fp32', 8, 7),
                                                     ('fp32', 32, 4), ('bf16', 128, 4)])
def test_early_target_forward_across_steps_is_bit_identical(env, precision, B, after_block):
    """Round 5: the target net's forward of step t + 1 (train.py:122) is enqueued on a stream of its own that does NOT wait for step t -- it
    reads the gathered next states (gathered on the upload stream), the target net's weight cache and nothing the learner is computing --
    and runs beside step t's backward pass.  A training loop shaped like the reference's (train.py:241-269: push one transition, sample from
    the HBM ring, train(), every third step copy the policy's weights into the target net) must give the same losses, TD targets,
    parameters and BatchNorm buffers BIT FOR BIT with the early stream on and off, on deterministic plans: the ring is pushed to and
    gathered from on one stream, the early stream waits for the target net's last weight change and for the last reader of the Q-map
    buffer it writes.  `after_block` = simq_plan_options.early_target_after_block: the point of step t's backward walk that forward is held
    back to (4 by default, -1 none, 7 / 0 the first / last residual block of the walk) -- ordering only.
    Round 6: ('fp32', 32, 4) and ('bf16', 128, 4) are the two schedules bench.py TIMES (configs[1] / configs[2]: ring-gathered minibatches of
    32 / 128, early stream on, block 4) -- a buffer-reuse race would depend on size and timing, so it is looked for at those sizes; and the
    switches are a StepOptions value of the call + an option of the ring, no longer module globals."""
    import random
    e, c = env, env['cases']
    sl = e['sl']
    cin, cout = 5, 2

    n0 = max(40, B + B // 4)                                                 # transitions in the ring before the first draw
    trs = e['synth'].make_transitions(n0 + 8, cin, cout, 11, terminal_frac=0.2)

    def loop(early):
        policy, target = _nets(e, precision, {'deterministic': 1, 'early_target_after_block': after_block}, cin, cout)
        assert policy.plan.options['early_target_after_block'] == after_block
        ring = e['simq'].DeviceReplayBuffer(max(64, 2 * B), cin, upload_stream=early)
        for t in trs[:n0]:
            ring.push(*t)
        random.seed(5)
        out = []
        for step in range(6):
            ring.push(*trs[n0 + step])                                       # the collector's hand-off (train.py:244)
            batch = ring.gather(ring.sample_indices(B))
            assert (getattr(batch, 'ready_event', None) is not None) == early
            info = sl.train_step(policy, target, batch, c.GAMMA, B, c.LR, c.MOMENTUM, c.WEIGHT_DECAY, c.CLIP, use_double_dqn=True,
                                 options=sl.StepOptions(early_target_forward=early))
            assert ('_qtgt_bufs' in policy.__dict__) == early                # (the early stream's two alternating Q-map buffers exist only then)
            out.append((info['loss'], info['td_error'], policy._last['y'].clone(), policy._last['q_sa'].clone()))
            if step % 3 == 2:
                target.copy_state_from(policy)                               # train.py:267-269
        torch.cuda.synchronize()
        return out, policy.flat_params.clone(), policy.bn_buffers.clone(), target.flat_params.clone()
    ref, p0, bn0, t0 = loop(False)
    got, p1, bn1, t1 = loop(True)
    for s, (a, b) in enumerate(zip(ref, got)):
        assert a[0] == b[0] and a[1] == b[1], 'step %d: loss / td error differ: %r vs %r' % (s, a[:2], b[:2])
        assert torch.equal(a[2], b[2]), 'step %d: TD targets differ (the target-net forward read stale next states or weights)' % s
        assert torch.equal(a[3], b[3]), 'step %d: q_sa differs' % s
    assert torch.equal(p0, p1) and torch.equal(bn0, bn1) and torch.equal(t0, t1)


def _group_nets(e, precision, seeds):
    """Two robot groups as train.py:180-195 builds them for lifting_2_pushing_2 (policies.py:35-42: Cout 2 and 1) + their rings."""
    groups = []
    for gi, cout in enumerate((2, 1)):
        cin = 5
        policy = e['simq'].FCN(cin, cout, precision=precision, options={'deterministic': 1})
        target = e['simq'].FCN(cin, cout, precision=precision, options={'deterministic': 1})
        policy.load_state_dict(e['ofcn'].state_from_numpy(e['synth'].make_state_dict(cin, cout, seeds[0] + gi)))
        target.load_state_dict(e['ofcn'].state_from_numpy(e['synth'].make_state_dict(cin, cout, seeds[1] + gi)))
        policy.train(); target.eval()
        ring = e['simq'].DeviceReplayBuffer(96, cin)
        for t in e['synth'].make_transitions(72, cin, cout, 31 + gi, terminal_frac=0.15):
            ring.push(*t)
        opt = torch.optim.SGD(policy.parameters(), lr=e['cases'].LR, momentum=e['cases'].MOMENTUM, weight_decay=e['cases'].WEIGHT_DECAY)
        groups.append(dict(policy=policy, target=target, ring=ring, opt=opt, cout=cout))
    return groups


@pytest.mark.parametrize('precision,B', [('fp32', 16), ('bf16', 32)])
def test_concurrent_robot_groups_equal_the_sequential_loop_bit_for_bit(env, precision, B):
    """Round 6 (train.py:253-261): the robot groups' Q-networks -- and the intention networks -- are independent, so simq.train_groups gives
    every learner a launch stream (+ side / early streams) of its own and enqueues all steps before it waits for a loss; the groups' steps
    then run side by side on the device.  Per net the same kernels run on the same operands in the same order, so on deterministic plans
    every net's losses, TD targets, parameters, momentum and BatchNorm buffers over a sample / train / target-sync loop must equal the
    reference's sequential loop (simq.train group after group on one stream) BIT FOR BIT -- including a DQNPolicy-style forward on the
    caller's stream right behind the pass (ordered behind the learner's own stream by FCN._order_behind_last_step, no join in the loop)
    and a target sync (train.py:267-269) issued from the caller's stream while the groups' streams may still be running."""
    import random
    import types
    e, c = env, env['cases']
    sl = e['sl']
    cfg = types.SimpleNamespace(batch_size=B, use_double_dqn=True, grad_norm_clipping=c.CLIP)
    probe = torch.from_numpy(e['synth'].make_states(2, 5, 91)[0][None]).cuda()

    def loop(concurrent):
        groups = _group_nets(e, precision, (3, 40))
        random.seed(17)
        log = []
        for step in range(5):
            batches = [g['ring'].sample(B) for g in groups]                                        # train.py:256
            if concurrent:
                infos = sl.train_groups(cfg, [g['policy'] for g in groups], [g['target'] for g in groups], [g['opt'] for g in groups],
                                        batches, None, [0.85, 0.85], concurrent=True)
                assert all(sl.learner_streams(g['policy']).launch is not None for g in groups)
            else:
                infos = [sl.train(cfg, g['policy'], g['target'], g['opt'], b, None, 0.85) for g, b in zip(groups, batches)]
            for g, info in zip(groups, infos):
                # policy.step's forward (policies.py:57-66, eval mode) on the CALLER's stream, right behind the pass
                g['policy'].eval()
                q = g['policy'].forward_nhwc(probe)
                g['policy'].train()
                log.append((info['loss'], info['td_error'], g['policy']._last['y'].clone(), g['policy']._last['q_sa'].clone(), q.clone()))
            if step % 2 == 1:
                for g in groups:
                    g['target'].load_state_dict(g['policy'].state_dict())                        # train.py:267-269
        torch.cuda.synchronize()
        fin = [(g['policy'].flat_params.clone(), g['policy'].bn_buffers.clone(), g['policy']._simq_opt_state.momentum.clone(),
                g['target'].flat_params.clone(), dict(g['policy'].num_batches_tracked)) for g in groups]
        return log, fin
    ref_log, ref_fin = loop(False)
    got_log, got_fin = loop(True)
    assert all(abs(a[0]) < 1e6 for a in ref_log)
    for i, (a, b) in enumerate(zip(ref_log, got_log)):
        assert a[0] == b[0] and a[1] == b[1], 'pass %d group %d: loss / td error differ: %r vs %r' % (i // 2, i % 2, a[:2], b[:2])
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), 'pass %d group %d: TD targets / q_sa differ' % (i // 2, i % 2)
        assert torch.equal(a[4], b[4]), 'pass %d group %d: the forward on the caller stream saw other parameters' % (i // 2, i % 2)
    for gi, (a, b) in enumerate(zip(ref_fin, got_fin)):
        assert all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4], 'group %d: final state differs' % gi


def test_concurrent_groups_with_intention_nets_equal_the_sequential_loop(env):
    """train.py:259-261 beside :255-257: the intention nets' supervision steps on launch streams of their own next to the robot groups' TD
    steps (simq.train_groups(intention_nets=...)): same losses and parameters as train() + train_intention() group after group."""
    import random
    import types
    e, c = env, env['cases']
    sl = e['sl']
    B = 8
    cfg = types.SimpleNamespace(batch_size=B, use_double_dqn=True, grad_norm_clipping=c.CLIP)

    def loop(concurrent):
        groups = _group_nets(e, 'fp32', (5, 60))
        inets, iopts = [], []
        for gi in range(2):
            net = e['simq'].FCN(4, 1, options={'deterministic': 1})
            net.load_state_dict(e['ofcn'].state_from_numpy(e['synth'].make_state_dict(4, 1, 70 + gi)))
            net.train()
            inets.append(net)
            iopts.append(torch.optim.SGD(net.parameters(), lr=c.LR, momentum=c.MOMENTUM, weight_decay=c.WEIGHT_DECAY))
        random.seed(23)
        log = []
        for step in range(3):
            batches = [g['ring'].sample(B) for g in groups]
            if concurrent:
                infos = sl.train_groups(cfg, [g['policy'] for g in groups], [g['target'] for g in groups], [g['opt'] for g in groups],
                                        batches, None, [0.85, 0.85], intention_nets=inets, optimizers_intention=iopts, concurrent=True)
            else:
                infos = []
                for g, b, net, opt in zip(groups, batches, inets, iopts):
                    info = sl.train(cfg, g['policy'], g['target'], g['opt'], b, None, 0.85)                  # train.py:257
                    info.update(sl.train_intention(net, opt, b, None))                                        # train.py:259-261
                    infos.append(info)
            log += [(i['loss'], i['td_error'], i['loss_intention']) for i in infos]
        torch.cuda.synchronize()
        return log, [n.flat_params.clone() for n in inets] + [g['policy'].flat_params.clone() for g in groups]
    ref_log, ref_p = loop(False)
    got_log, got_p = loop(True)
    assert ref_log == got_log, (ref_log, got_log)
    assert all(torch.equal(a, b) for a, b in zip(ref_p, got_p))
