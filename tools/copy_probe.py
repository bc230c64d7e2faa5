#!/usr/bin/env python3
"""GPU (under rocprofv3 --kernel-trace): where do the ~69 small __amd_rocclr_copyBuffer launches per bench step come from?
mode 'fixed': the same gathered batch every step (no ring gather); mode 'gather': one ring gather per step, no training."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np, torch
import simq
from simq import synth
from simq.learner import _opt_state, train_step
mode = sys.argv[1]
dev = torch.device('cuda', 0)
cin, cout, B = 4, 2, 32
policy, target = simq.FCN(cin, cout, device=dev), simq.FCN(cin, cout, device=dev)
target.copy_state_from(policy); policy.train(); target.eval()
trs = synth.make_transitions(256, cin, cout, 5, terminal_frac=0.1)
ring = simq.DeviceReplayBuffer(256, cin, device=dev)
ring.push_many(np.stack([t[0] for t in trs]), [t[1] for t in trs], [t[2] for t in trs],
               np.stack([t[3] if t[3] is not None else np.zeros_like(t[0]) for t in trs]), [t[3] is None for t in trs])
random.seed(1)
batch = ring.gather(ring.sample_indices(B))
st = _opt_state(policy, None)
torch.cuda.synchronize()
for _ in range(10):
    if mode == 'fixed':
        out = train_step(policy, target, batch, 0.75, B, 0.01, 0.9, 1e-4, 100.0, opt_state=st, sync=False)
        out.tolist()
    else:
        batch = ring.gather(ring.sample_indices(B))
        torch.cuda.synchronize()
