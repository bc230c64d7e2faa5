#!/usr/bin/env python3
"""Diagnostic (GPU box): loss / TD error of one configs[2]-shaped step (Cin 5, B = 128) in bf16 and exact fp32 for a few seeds -- how far
the bf16 plan's scalar loss sits from fp32's, and how much of that moves with the summation order alone (run once with the product
library and once with SIMQ_LIBRARY=libsimq_ablate.so SIMQ_BF16_IMG_HALF=0, which routes the 128-channel layers through the whole-map
kernels of round 3).  Backs the bars of tests/test_gpu_fullsize.py::test_b128_bf16_config_runs_and_is_consistent."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import simq  # noqa: E402
from oracle import cases, fcn as ofcn  # noqa: E402
from simq import synth  # noqa: E402
from simq.learner import Transition, train_step  # noqa: E402


def step(cin, cout, trs, B, seed, precision):
    policy, target = simq.FCN(cin, cout, precision=precision), simq.FCN(cin, cout, precision=precision)
    policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed)))
    target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, seed + 1)))
    policy.train()
    target.eval()
    info = train_step(policy, target, Transition(*zip(*trs)), cases.GAMMA, B, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, cases.CLIP)
    return policy, info


print('library', os.environ.get('SIMQ_LIBRARY', 'product'), 'SIMQ_BF16_IMG_HALF', os.environ.get('SIMQ_BF16_IMG_HALF'))
for tseed, wseed in ((31, 13), (32, 14), (33, 15), (34, 16), (35, 17), (36, 18)):
    trs = synth.make_transitions(128, 5, 2, tseed, terminal_frac=0.1)
    pa, ia = step(5, 2, trs, 128, wseed, 'bf16')
    pf, i32 = step(5, 2, trs, 128, wseed, 'fp32')
    dq = float((pa._last['q_sa'] - pf._last['q_sa']).abs().max() / pf._last['q_sa'].abs().max())
    print('seeds (%d, %d): loss bf16 %.4f fp32 %.4f rel %.4f | td bf16 %.4f fp32 %.4f rel %.4f | q_sa max rel %.4f' % (
        tseed, wseed, ia['loss'], i32['loss'], abs(ia['loss'] - i32['loss']) / abs(i32['loss']),
        ia['td_error'], i32['td_error'], abs(ia['td_error'] - i32['td_error']) / abs(i32['td_error']), dq), flush=True)
