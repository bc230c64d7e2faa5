#!/usr/bin/env python3
"""GPU: repeated train() steps on ONE fixed minibatch (target net fixed), fp32 vs bf16 plans from the same initial weights -- does the
TD loss fall the same way?   usage: tests/diag/overfit_check.py [steps] [B]   (PRECS=fp32,bf16,bf16x3)
Finding (round 2): the problem is chaotic -- momentum 0.9, double-DQN argmax flips, TD errors of O(1) per transition.  90 steps at B = 16:
HIP fp32 ends at 0.33, the reference's own fp32 modules on the CPU at 3.5 (mean of the last 20 steps 4.5), the reference under
torch.autocast(bf16) at 7.1, HIP bf16x3 at 1.4, HIP bf16 between 1.6 and 18 from run to run (atomics order).  Trajectories separate
after ~10 steps whatever the arithmetic, so this is a smoke check for finiteness, not a parity measure; tests/golden G8 is."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np, torch
import simq
from simq import synth
from oracle import cases, fcn as ofcn, learner as olearner

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cin, cout = 4, 2
out = {}
for prec in (os.environ.get('PRECS', 'fp32,bf16').split(',')):
    policy, target = simq.FCN(cin, cout, precision=prec), simq.FCN(cin, cout, precision=prec)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 3))); policy.train(True)
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 1003))); target.train(False)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    batch = cases.make_batch(cin, cout, B, 11)
    cfg = cases.make_cfg(B)
    losses = []
    for s in range(steps):
        info = simq.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
        losses.append(info['loss'])
    out[prec] = losses
    print(prec, ' '.join('%.4g' % v for v in losses[::max(1, steps // 15)]), '| last', '%.4g' % losses[-1])
