#!/usr/bin/env python3
"""GPU smoke: one train() step for input channel counts 10 / 9 / 3 / 1 in bf16 and fp32 (Cin = 10 is outside the bf16 stem kernel's
7 * Cin <= 63 and falls back to the fp32 stem; the others take it).  Prints loss / td error and whether the parameters stayed finite."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch, numpy as np
import simq
from simq import synth
from oracle import cases, fcn as ofcn, learner as olearner
for cin in (10, 9, 3, 1):
    for prec in ('bf16', 'fp32'):
        policy, target = simq.FCN(cin, 2, precision=prec), simq.FCN(cin, 2, precision=prec)
        policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, 2, 3))); policy.train(True)
        target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, 2, 1003))); target.train(False)
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        info = simq.train(cases.make_cfg(8), policy, target, opt, cases.make_batch(cin, 2, 8, 11), olearner.apply_transform, cases.GAMMA)
        print(cin, prec, info, bool(torch.isfinite(policy.flat_params).all()))
