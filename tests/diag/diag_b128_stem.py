"""Diagnostic (GPU box): the sampled elements of the stem weight gradient of gs_b128_09 -- HIP against the fp64 oracle, element by element."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import simq  # noqa: E402
from oracle import cases  # noqa: E402
from oracle import learner as olearner  # noqa: E402
import test_gpu_fcn as T  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'grad_study_b128.npz'))
for cname in sys.argv[1:] or ['gs_b128_09', 'gs_b128_03']:
    name, cin, cout, B, wseed, dseed = [c for c in cases.GRAD_STUDY_B128_CASES if c[0] == cname][0]
    cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
    policy, target = T.make_net(simq, cin, cout, wseed, True), T.make_net(simq, cin, cout, wseed + 1000, False)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    simq.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    tn = float(policy._simq_opt_state.total_norm.item())
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    grads = [v.detach().cpu().double() / coef for v in policy.reference_views(policy.flat_grads)]
    ref = g[name + '.grad64']
    print(name, 'total norm', tn, 'clip coef', coef)
    for ti in (0, 1, 2, 3, 4):
        t = grads[ti]
        hs = t.reshape(-1)[torch.tensor(cases.sample_indices(t.numel()))].numpy()
        print(' tensor', ti, tuple(t.shape), 'rel err', float(np.sqrt(((hs - ref[ti]) ** 2).sum() / (ref[ti] ** 2).sum())))
        print('   hip ', np.array2string(hs[:8], precision=5))
        print('   fp64', np.array2string(ref[ti][:8], precision=5))
        print('   diff', np.array2string((hs - ref[ti])[:8], precision=5))
    # the stem weight gradient per input channel: error norm by ci (OIHW view [64, Cin, 7, 7])


# the same gradient through the DENSE backward (autograd path: torch's Huber + gather produce a dense dLoss/dQ, simq_backward walks it): a
# different head arithmetic in front of the identical network -- if the error of the one-hot form is luck, this one lands elsewhere
import torch.nn.functional as F  # noqa: E402
rl2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
for cname in sys.argv[1:] or ['gs_b128_09', 'gs_b128_03', 'gs_b128_00']:
    name, cin, cout, B, wseed, dseed = [c for c in cases.GRAD_STUDY_B128_CASES if c[0] == cname][0]
    cfg, batch = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed)
    policy, target = T.make_net(simq, cin, cout, wseed, True), T.make_net(simq, cin, cout, wseed + 1000, False)
    opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    simq.train(cfg, policy, target, opt, batch, olearner.apply_transform, cases.GAMMA)
    y = policy._last['y'].clone()
    tn = float(policy._simq_opt_state.total_norm.item())
    coef = min(1.0, cases.CLIP / (tn + 1e-6))
    g1 = [v.detach().cpu().double() / coef for v in policy.reference_views(policy.flat_grads)]
    policy2 = T.make_net(simq, cin, cout, wseed, True)
    x = torch.cat([olearner.apply_transform(s) for s in batch.state]).cuda()
    act = torch.tensor(batch.action, dtype=torch.long, device='cuda')
    q = policy2(x)
    loss = F.smooth_l1_loss(q.view(B, -1).gather(1, act.unsqueeze(1)).squeeze(1), y)
    loss.backward()
    g2 = [p.grad.detach().cpu().double() for p in policy2.parameters() if p.grad is not None]
    ref = g[name + '.grad64']
    samp = lambda gs: np.stack([t.reshape(-1)[torch.tensor(cases.sample_indices(t.numel()))].numpy() for t in gs])
    print('%s: one-hot backward %.3g | dense backward (autograd path) %.3g | between the two %.3g | reference fp32 %.3g'
          % (name, rl2(samp(g1), ref), rl2(samp(g2), ref), rl2(samp(g1), samp(g2)), float(g[name + '.ref_grad_err'])), flush=True)
