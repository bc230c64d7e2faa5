#!/usr/bin/env python3
"""GPU: latency of DQNPolicy.step (policies.py:47-74, batch-1 eval forward + argmax) and of a batched eval forward."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch
import simq
from simq import synth
for prec in ('fp32', 'bf16'):
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 1}], num_input_channels=4, final_exploration=0.01, checkpoint_path=None, simq_precision=prec)
    pol = simq.DQNPolicy(cfg, train=False, random_seed=0)
    s = synth.make_states(9, 4, 3)
    for _ in range(5): pol.step([[s[0]]], exploration_eps=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 50
    for i in range(N): a = pol.step([[s[i % 9]]], exploration_eps=0.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    net = pol.policy_nets[0]
    x8 = torch.from_numpy(s[:8]).cuda()
    with torch.no_grad():
        for _ in range(3): net.forward_nhwc(x8)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net.forward_nhwc(x8)
        torch.cuda.synchronize(); dt8 = (time.perf_counter() - t0) / 20
    print('%s: DQNPolicy.step (B=1, incl. H2D + argmax + D2H of the Q-map) %.3f ms ; eval forward B=8 %.3f ms (%.3f ms/state)' % (prec, dt * 1e3, dt8 * 1e3, dt8 * 1e3 / 8))
