"""Diagnostic (GPU box, not collected by pytest): what F(4x4,3x3) in the GRAD-MODE forward of the widest layers does to the gradient's
error.  Runs the gradient study of tests/test_gpu_fcn.py::test_gradient_parity_distribution (13 batches of 8 / 32, six of 64 and twelve of 32
against the fp64 oracle's gradient, with the reference's own fp32 error on the same elements) for several values of
simq_plan_options.winograd_f4_fwd_grad_min_cc: 0 = F(2x2,3x3) everywhere (default), 262144 = the 512->512 convolutions, 131072 = + 256->512,
65536 = + layer3, 16384 = every Winograd layer.   usage: python tests/diag_f4_grad_layers.py [threshold ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import simq  # noqa: E402
from simq import _lib  # noqa: E402
from oracle import cases  # noqa: E402
from oracle import learner as olearner  # noqa: E402
import test_gpu_fcn as T  # noqa: E402

rl2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
for thr in [int(a) for a in sys.argv[1:]] or [0, 262144, 131072, 65536, 16384]:
    _lib.DEFAULT_PLAN_OPTIONS.clear()
    _lib.DEFAULT_PLAN_OPTIONS['winograd_f4_fwd_grad_min_cc'] = thr
    for fixture, case_list in (('grad_study.npz', cases.GRAD_STUDY_CASES), ('grad_study_b64.npz', cases.GRAD_STUDY_B64_CASES),
                               ('grad_study_b32.npz', cases.GRAD_STUDY_B32_CASES)):
        g = np.load(os.path.join(ROOT, 'tests', 'golden', fixture))
        rows = []
        for name, cin, cout, B, wseed, dseed in case_list:
            cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
            policy, target = T.make_net(simq, cin, cout, wseed, True), T.make_net(simq, cin, cout, wseed + 1000, False)
            opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
            info1 = simq.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
            tn = float(policy._simq_opt_state.total_norm.item())
            coef = min(1.0, cases.CLIP / (tn + 1e-6))
            grads = [v.detach().cpu().double() / coef for v in policy.reference_views(policy.flat_grads)]
            gs = [t.reshape(-1)[torch.tensor(cases.sample_indices(t.numel()))].numpy() for t in grads]
            rows.append((rl2(np.stack(gs), g[name + '.grad64']), float(g[name + '.ref_grad_err']),
                         abs(info1['loss'] - float(g[name + '.loss64'][0])) / float(g[name + '.loss64'][0])))
            del policy, target, opt
        e, r, l = np.array(rows).T
        print('min_cc %7d  %-18s gradient error: median %.3g (reference fp32 %.3g, ratio %.2f)  mean %.3g (%.3g)  worst case / its reference %.2f  | loss step 1 max %.1e'
              % (thr, fixture, np.median(e), np.median(r), np.median(e) / np.median(r), e.mean(), r.mean(), (e / np.maximum(r, np.median(r))).max(), l.max()), flush=True)
