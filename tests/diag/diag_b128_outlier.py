"""Diagnostic (GPU box): the B = 128 gradient study's per-case errors under the plan options that change the arithmetic of the differentiated path --
where does an outlier (gs_b128_09: 2.0e-2 against the reference's 2.9e-3) come from?  usage: python tests/diag/diag_b128_outlier.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import simq  # noqa: E402
from simq import _lib  # noqa: E402
from oracle import cases  # noqa: E402
from oracle import learner as olearner  # noqa: E402
import test_gpu_fcn as T  # noqa: E402

rl2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'grad_study_b128.npz'))
variants = [('default', {}), ('grad fwd all F(2x2)', {'winograd_f4_fwd_grad_min_cc': 0}), ('+ dgrads F(2x2)', {'winograd_f4_fwd_grad_min_cc': 0, 'winograd_f4_grad': 0}),
            ('+ wgrad F(2x2)', {'winograd_f4_fwd_grad_min_cc': 0, 'winograd_f4_grad': 0, 'winograd_wgrad_f4': 0}), ('no Winograd', {'winograd': 0}),
            ('no BN1 fusion', {'fuse_bn1_apply': 0}), ('stem sums unfused', {'fuse_stem_backward_sums': 0}), ('all BN sums unfused', {'fuse_bn_backward_sums': 0, 'fuse_bn1_apply': 0}),
            ('deterministic', {'deterministic': 1})]
which = [c for c in cases.GRAD_STUDY_B128_CASES if c[0] in ('gs_b128_09', 'gs_b128_03')]
for vname, opts in variants:
    _lib.DEFAULT_PLAN_OPTIONS.clear()
    _lib.DEFAULT_PLAN_OPTIONS.update(opts)
    row = []
    for name, cin, cout, B, wseed, dseed in which:
        cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
        policy, target = T.make_net(simq, cin, cout, wseed, True), T.make_net(simq, cin, cout, wseed + 1000, False)
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        simq.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
        tn = float(policy._simq_opt_state.total_norm.item())
        coef = min(1.0, cases.CLIP / (tn + 1e-6))
        grads = [v.detach().cpu().double() / coef for v in policy.reference_views(policy.flat_grads)]
        gs = np.stack([t.reshape(-1)[torch.tensor(cases.sample_indices(t.numel()))].numpy() for t in grads])
        ref = g[name + '.grad64']
        per_tensor = np.sqrt(((gs - ref) ** 2).sum(1))
        worst = int(np.argmax(per_tensor))
        row.append('%s %.3g (ref %.3g; largest share: tensor %d, %.0f %% of the error)' % (name, rl2(gs, ref), float(g[name + '.ref_grad_err']), worst,
                                                                                       100 * per_tensor[worst] ** 2 / (per_tensor ** 2).sum()))
        del policy, target, opt
    print('%-22s %s' % (vname, ' | '.join(row)), flush=True)
