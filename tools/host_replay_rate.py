#!/usr/bin/env python3
"""GPU: the PCIe-inclusive rate of the drop-in path -- the reference's own host ReplayBuffer (a python list of numpy transitions,
train.py:28-45) sampled every step and handed to simq.train() as host tuples (train.py:252-258), so every step stacks and uploads its
2 x B x 96 x 96 x C states.  bench.py's `value` keeps the replay ring in HBM instead; this is the number to quote beside it.
usage: host_replay_rate.py [fp32|bf16] [B] [Cin]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np, torch
import simq
from simq import synth
from types import SimpleNamespace
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device('cuda', 0)
torch.manual_seed(1)
policy, target = simq.FCN(cin, 2, device=dev, precision=prec), simq.FCN(cin, 2, device=dev, precision=prec)
target.copy_state_from(policy); policy.train(); target.eval()
opt = torch.optim.SGD(policy.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
cfg = SimpleNamespace(batch_size=B, use_double_dqn=True, grad_norm_clipping=100)
buf = simq.ReplayBuffer(2000)
for t in synth.make_transitions(2000, cin, 2, 5, terminal_frac=0.1):
    buf.push(*t)
random.seed(3)
def step():
    return simq.train(cfg, policy, target, opt, buf.sample(B), None, 0.75)
for _ in range(3):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    info = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('host replay (PCIe inclusive) %s B=%d Cin=%d: %.1f transitions/s, %.3f ms per step, loss %.4f' % (prec, B, cin, B * n / dt, dt / n * 1e3, info['loss']))
