#!/usr/bin/env python3
"""GPU box: the full TD step of one bench workload on the ABLATION build (libsimq_ablate.so), so that SIMQ_* kernel-selection switches
can be A/B-ed at the level that counts -- bench.py itself runs the product library only.  One process = one setting (the switches are
read once); alternate the settings from a shell loop and compare on the same box:
    for i in 1 2 3; do SIMQ_IMG_F32=0 python tools/ab_step.py; SIMQ_IMG_F32=1 python tools/ab_step.py; done
Plan options (simq_plan_options fields) can be A/B-ed the same way: SIMQ_AB_OPTIONS='{"winograd_min_cc": 8192}'.
usage: ab_step.py [configs1|configs2] [steps]   -> one line: transitions/s, ms per step"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
os.environ.setdefault('SIMQ_LIBRARY', os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
import torch  # noqa: E402
import simq  # noqa: E402
from simq import synth  # noqa: E402
from simq.learner import train_step  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'configs1'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cin, cout, B, precision = {'configs1': (4, 2, 32, 'fp32'), 'configs2': (5, 2, 128, 'bf16')}[workload]
dev = torch.device('cuda:0')
options = json.loads(os.environ.get('SIMQ_AB_OPTIONS', '{}')) or None
torch.manual_seed(20260928)
policy = simq.FCN(cin, cout, device=dev, precision=precision, options=options)
target = simq.FCN(cin, cout, device=dev, precision=precision, options=options)
target.copy_state_from(policy)
policy.train()
target.eval()
opt = simq.learner._opt_state(policy, None)
ring = simq.DeviceReplayBuffer(1024, cin, device=dev)
for t in synth.make_transitions(1024, cin, cout, 5, terminal_frac=0.1):
    ring.push(*t)


def step():
    batch = ring.gather(ring.sample_indices(B))
    return train_step(policy, target, batch, 0.75, B, 0.01, 0.9, 1e-4, 100.0, use_double_dqn=True, opt_state=opt)


for _ in range(5):
    info = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    info = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tag = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('SIMQ_') and k not in ('SIMQ_LIBRARY',))
print('%s %s: %.1f tr/s  %.3f ms/step  loss %.4f  [%s]' % (workload, precision, B * steps / dt, dt / steps * 1e3, info['loss'], tag))
