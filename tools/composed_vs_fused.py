#!/usr/bin/env python3
"""GPU aid: throughput of the single-call library step (simq_train_step) vs the same launches issued from Python (the form the
data-parallel path uses), BASELINE configs[1] workload."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np, torch
import simq
import simq.learner as sl
from simq import synth
torch.manual_seed(20260928)
policy, target = simq.FCN(4, 2), simq.FCN(4, 2)
target.copy_state_from(policy); policy.train(); target.eval()
st = sl._opt_state(policy, None)
trs = synth.make_transitions(1024, 4, 2, 5, terminal_frac=0.1)
ring = simq.DeviceReplayBuffer(1024, 4)
ring.push_many(np.stack([t[0] for t in trs]), [t[1] for t in trs], [t[2] for t in trs],
               np.stack([t[3] if t[3] is not None else np.zeros_like(t[0]) for t in trs]), [t[3] is None for t in trs])
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        sl.train_step(policy, target, ring.gather(ring.sample_indices(32)), 0.75, 32, 0.01, 0.9, 1e-4, 100.0, opt_state=st)
    torch.cuda.synchronize(); return 32 * n / (time.perf_counter() - t0)
random.seed(1)
for fused in (True, False, True, False):
    sl.FUSED_LIBRARY_STEP = fused
    run(3)
    print('%-28s %.1f tr/s' % ('library step (one C call):' if fused else 'composed from Python:', run(20)))
