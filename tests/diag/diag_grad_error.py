"""Diagnostic (GPU): per-tensor gradient error of the HIP path and of the fp32 oracle, both vs the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
import simq
from simq import synth
from oracle import cases, fcn as ofcn, learner as ol

name, cin, cout, B, wseed, dseed = cases.TRAIN_CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
PREC = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
cfg = cases.make_cfg(B); batch = cases.make_batch(cin, cout, B, dseed); spec = ofcn.state_spec(cin, cout)
ex = {}
for dt in (torch.float32, torch.float64):
    st, tg = cases.oracle_state(cin, cout, wseed, dt), cases.oracle_state(cin, cout, wseed + 1000, dt)
    e = {}
    ol.train_step(cfg, st, tg, spec, [None] * len(ol.grad_keys(spec)), batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, dtype=dt, extras=e)
    ex[dt] = e
policy = simq.FCN(cin, cout, precision=PREC); policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, wseed))); policy.train()
target = simq.FCN(cin, cout, precision=PREC); target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, wseed + 1000))); target.eval()
opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
simq.train(cfg, policy, target, opt, batch, None, cases.GAMMA)
tn = float(policy._simq_opt_state.total_norm.item()); coef = min(1.0, cases.CLIP / (tn + 1e-6))
g = policy.flat_grads.cpu() / coef
print('total_norm hip %.6f  o32 %.6f  o64 %.6f' % (tn, ex[torch.float32]['total_norm'], ex[torch.float64]['total_norm']))
print('q err vs o64: hip %.3g  o32 %.3g' % (float((policy._last['q'].cpu().double() - ex[torch.float64]['output']).abs().max() / ex[torch.float64]['output'].abs().max()),
      float((ex[torch.float32]['output'].double() - ex[torch.float64]['output']).abs().max() / ex[torch.float64]['output'].abs().max())))
for (nm, _, kind), (off, n, shape) in zip(policy._param_names, policy._grad_views):
    t = g[off:off + n].view(shape)
    if len(shape) == 4: t = t.permute(0, 3, 1, 2)
    r64 = ex[torch.float64]['grads']['module.' + nm]; r32 = ex[torch.float32]['grads']['module.' + nm].double()
    d = float(r64.norm())
    print('%-40s |g|=%.3e  hip %.2e  o32 %.2e' % (nm, d, float((t.double() - r64).norm()) / max(d, 1e-30), float((r32 - r64).norm()) / max(d, 1e-30)))
