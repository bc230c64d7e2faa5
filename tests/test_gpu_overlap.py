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
    for rep in range(12):                           # (12 plans, each creating its side stream and 4 events; a leak shows in rocm-smi, a crash here)
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


@pytest.mark.parametrize('precision,B,after_block', [('fp32', 8, 4), ('bf16', 16, 4), ('fp32', 8, -1), ('bf16', 16, 0), ('fp32', 8, 7)])
def test_early_target_forward_across_steps_is_bit_identical(env, precision, B, after_block):
    """Round 5: the target net's forward of step t + 1 (train.py:122) is enqueued on a stream of its own that does NOT wait for step t -- it
    reads the gathered next states (gathered on the upload stream), the target net's weight cache and nothing the learner is computing --
    and runs beside step t's backward pass.  A training loop shaped like the reference's (train.py:241-269: push one transition, sample from
    the HBM ring, train(), every third step copy the policy's weights into the target net) must give the same losses, TD targets,
    parameters and BatchNorm buffers BIT FOR BIT with the early stream on and off, on deterministic plans: the ring is pushed to and
    gathered from on one stream, the early stream waits for the target net's last weight change and for the last reader of the Q-map
    buffer it writes.  `after_block` = simq_plan_options.early_target_after_block: the point of step t's backward walk that forward is held
    back to (4 by default, -1 none, 7 / 0 the first / last residual block of the walk) -- ordering only."""
    import random
    e, c = env, env['cases']
    sl = e['sl']
    cin, cout = 5, 2

    def loop(early):
        keep = (sl.EARLY_TARGET_FORWARD, sl.GATHER_ON_UPLOAD_STREAM)
        sl.EARLY_TARGET_FORWARD, sl.GATHER_ON_UPLOAD_STREAM = early, early
        try:
            policy, target = _nets(e, precision, {'deterministic': 1, 'early_target_after_block': after_block}, cin, cout)
            assert policy.plan.options['early_target_after_block'] == after_block
            ring = e['simq'].DeviceReplayBuffer(64, cin)
            trs = e['synth'].make_transitions(48, cin, cout, 11, terminal_frac=0.2)
            for t in trs[:40]:
                ring.push(*t)
            random.seed(5)
            out = []
            for step in range(6):
                ring.push(*trs[40 + step])                                   # the collector's hand-off (train.py:244)
                batch = ring.gather(ring.sample_indices(B))
                assert (getattr(batch, 'ready_event', None) is not None) == early
                info = sl.train_step(policy, target, batch, c.GAMMA, B, c.LR, c.MOMENTUM, c.WEIGHT_DECAY, c.CLIP, use_double_dqn=True)
                out.append((info['loss'], info['td_error'], policy._last['y'].clone(), policy._last['q_sa'].clone()))
                if step % 3 == 2:
                    target.copy_state_from(policy)                           # train.py:267-269
            torch.cuda.synchronize()
            return out, policy.flat_params.clone(), policy.bn_buffers.clone(), target.flat_params.clone()
        finally:
            sl.EARLY_TARGET_FORWARD, sl.GATHER_ON_UPLOAD_STREAM = keep
    ref, p0, bn0, t0 = loop(False)
    got, p1, bn1, t1 = loop(True)
    for s, (a, b) in enumerate(zip(ref, got)):
        assert a[0] == b[0] and a[1] == b[1], 'step %d: loss / td error differ: %r vs %r' % (s, a[:2], b[:2])
        assert torch.equal(a[2], b[2]), 'step %d: TD targets differ (the target-net forward read stale next states or weights)' % s
        assert torch.equal(a[3], b[3]), 'step %d: q_sa differs' % s
    assert torch.equal(p0, p1) and torch.equal(bn0, bn1) and torch.equal(t0, t1)
