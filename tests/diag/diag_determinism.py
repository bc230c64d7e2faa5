#!/usr/bin/env python
"""Which results of one TD step differ between two runs from identical state (GPU box)?  Runs the same seeded step N times in one process
and compares Q-map, TD targets, loss sums, BatchNorm buffers and every gradient tensor BITWISE against the first run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
import simq  # noqa: E402
import simq.learner as sl  # noqa: E402
from oracle import cases, fcn as ofcn  # noqa: E402
from simq import synth  # noqa: E402


def run(precision, B, options, cin=5, cout=2):
    policy, target = simq.FCN(cin, cout, precision=precision, options=options), simq.FCN(cin, cout, precision=precision, options=options)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 3)))
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 4)))
    policy.train(); target.eval()
    info = sl.train_step(policy, target, cases.make_batch(cin, cout, B, 7), cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP, use_double_dqn=True)
    torch.cuda.synchronize()
    names = [n for n, _ in policy.named_parameters()]
    return dict(loss=info['loss'], q=policy._last['q'].clone(), y=policy._last['y'].clone(), bn=policy.bn_buffers.clone(),
                grads=[v.clone() for v in policy.reference_views(policy.flat_grads)], names=names, params=policy.flat_params.clone())


for precision, B in (('fp32', 32), ('bf16', 64)):
    for opts in ({}, {'deterministic': 1}):
        try:
            runs = [run(precision, B, opts) for _ in range(4)]
        except Exception as ex:      # noqa: BLE001  (option unknown in this build)
            print(precision, opts, 'skipped:', ex)
            continue
        a = runs[0]
        for i, b in enumerate(runs[1:], 1):
            diff = [n for n, x, y in zip(a['names'], a['grads'], b['grads']) if not torch.equal(x, y)]
            print('%s B=%d %s run %d vs 0: loss %s, Q %s, y %s, BN buffers %s, params %s, gradient tensors differing: %d of %d %s'
                  % (precision, B, opts, i, a['loss'] == b['loss'], torch.equal(a['q'], b['q']), torch.equal(a['y'], b['y']), torch.equal(a['bn'], b['bn']),
                     torch.equal(a['params'], b['params']), len(diff), len(a['names']), diff))
