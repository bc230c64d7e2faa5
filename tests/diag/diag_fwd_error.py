"""Diagnostic (GPU): per-layer forward error of a precision mode vs the fp64 oracle (and the fp32 oracle's own error)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
import simq
from simq import synth, _lib
_lib.DEFAULT_PLAN_OPTIONS['keep_fp32_activations'] = 1   # this tool reads the fp32 copies of the block activations (FCN.saved_activation)
from oracle import cases, fcn as ofcn, learner as ol

PREC = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
TRAIN = (sys.argv[2] == 'train') if len(sys.argv) > 2 else False
cin, cout, B, wseed, dseed = 4, 2, 2, 11, 21
x_hwc = synth.make_states(B, cin, dseed)
x = torch.cat([ol.apply_transform(s) for s in x_hwc])
taps = {}
for dt in (torch.float32, torch.float64):
    st = cases.oracle_state(cin, cout, wseed, dt); t = {}
    with torch.no_grad():
        q = ofcn.fcn_forward(st, x.to(dt), TRAIN, t)
    t['q'] = q; taps[dt] = t
net = simq.FCN(cin, cout, precision=PREC); net.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(cin, cout, wseed))); net.train(TRAIN)
with torch.no_grad():
    q = net.forward_nhwc(torch.from_numpy(x_hwc).cuda())
def err(a, b): return float((a.double() - b).abs().max() / b.abs().max())
def l2(a, b): return float((a.double() - b).norm() / b.norm())
print('precision', PREC, 'train' if TRAIN else 'eval')
# (head.a1 / head.a2: a no-grad forward does not store them -- the eval head is folded into one pass, fp32 plans apply the head's
#  BatchNorm 1 inside conv2's operand staging: include/simq.h simq_workspace_tensor)
for name in ['stem.pool'] + ['layer%d.%d' % (l, b) for l in range(1, 5) for b in range(2)]:
    h = net.saved_activation(name, B).cpu().permute(0, 3, 1, 2)
    r64, r32 = taps[torch.float64][name], taps[torch.float32][name]
    print('%-10s max-norm err: hip %.2e  o32 %.2e | rel-L2: hip %.2e  o32 %.2e' % (name, err(h, r64), err(r32, r64), l2(h, r64), l2(r32, r64)))
print('%-10s max-norm err: hip %.2e  o32 %.2e | rel-L2: hip %.2e  o32 %.2e' % ('q', err(q.cpu(), taps[torch.float64]['q']), err(taps[torch.float32]['q'], taps[torch.float64]['q']), l2(q.cpu(), taps[torch.float64]['q']), l2(taps[torch.float32]['q'], taps[torch.float64]['q'])))
