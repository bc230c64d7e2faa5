#!/usr/bin/env python
"""GPU box: seeded TD steps of a DETERMINISTIC fp32 plan under every simq_tune_wgrad_overlap / simq_tune_fwd_overlap setting -- the weight gradients behind the
dgrads (0), beside them (1, 3) and one block behind on a second set of temporaries (4) run the same kernels on the same operands, so the
loss, the gradient and the parameters after the step must be equal BIT FOR BIT.  Also two consecutive steps (the second set's events are
re-recorded) and a default (atomics) plan at 1e-5."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
import simq  # noqa: E402
import simq.learner as sl  # noqa: E402
from oracle import cases, fcn as ofcn  # noqa: E402
from simq import synth  # noqa: E402
from simq._lib import lib  # noqa: E402


def run(mode, B, options, steps=2, cin=5, cout=2, fwd=2):
    lib.call('simq_tune_wgrad_overlap', mode)
    lib.call('simq_tune_fwd_overlap', fwd)
    policy, target = simq.FCN(cin, cout, precision='fp32', options=options), simq.FCN(cin, cout, precision='fp32', options=options)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 3)))
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 4)))
    policy.train(); target.eval()
    losses = []
    for s in range(steps):
        info = sl.train_step(policy, target, cases.make_batch(cin, cout, B, 7 + s), cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                             cases.CLIP, use_double_dqn=True)
        losses.append(info['loss'])
    torch.cuda.synchronize()
    return dict(loss=losses, grads=policy.flat_grads.clone(), params=policy.flat_params.clone(), bn=policy.bn_buffers.clone())


bad = 0
for B in (8, 32):
    ref = run(0, B, {'deterministic': 1})
    for mode in (1, 3, 4):
        r = run(mode, B, {'deterministic': 1})
        same = r['loss'] == ref['loss'] and all(torch.equal(r[k], ref[k]) for k in ('grads', 'params', 'bn'))
        print('deterministic fp32 B=%d wgrad_overlap=%d vs 0: bit-identical %s (loss %s)' % (B, mode, same, r['loss']))
        bad += 0 if same else 1
    for fwd in (0, 1):         # the fork point of the no-grad forwards: the deferred running-statistics update must reproduce the serial order
        r = run(0, B, {'deterministic': 1}, fwd=fwd)
        same = r['loss'] == ref['loss'] and all(torch.equal(r[k], ref[k]) for k in ('grads', 'params', 'bn'))
        print('deterministic fp32 B=%d fwd_overlap=%d vs 2: bit-identical %s (BatchNorm buffers %s)' % (B, fwd, same, torch.equal(r['bn'], ref['bn'])))
        bad += 0 if same else 1
    ref = run(0, B, {}, fwd=0)
    r = run(4, B, {})
    rel = float((r['grads'] - ref['grads']).norm() / ref['grads'].norm())
    # (a default plan sums its weight gradients with fp32 atomics: 2e-3 ... 6e-3 between ANY two runs; informational)
    print('default fp32 B=%d wgrad_overlap=4 fwd_overlap=2 vs 0 / 0: gradient rel-L2 %.2e, params max diff %.2e, BatchNorm buffers equal %s'
          % (B, rel, float((r['params'] - ref['params']).abs().max()), torch.equal(r['bn'], ref['bn'])))
    bad += 0 if rel < 5e-2 else 1
lib.call('simq_tune_wgrad_overlap', 1)
lib.call('simq_tune_fwd_overlap', 2)
print('WOV_CHECK', 'FAIL' if bad else 'PASS')
sys.exit(1 if bad else 0)
