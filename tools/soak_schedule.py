#!/usr/bin/env python3
"""GPU box: a long training loop shaped like train.py:241-269 (push one transition, sample from the HBM ring, train(), copy the policy's weights
into the target net every `sync_every` steps) run TWICE on deterministic plans -- once with the bench's schedule (three forwards side by side,
piped weight gradients, upload stream, early target forward held to block 4, the loss read without waiting for the backward pass) and once
fully serial (fwd_overlap = wgrad_overlap = 0, no upload stream, no early forward) -- and compared BIT FOR BIT after every step.  The suite's
tests/test_gpu_overlap.py does this for 2-6 steps; a buffer-reuse or ordering race that needs an unlucky interleaving shows up over hundreds.
usage: soak_schedule.py [fp32|bf16] [batch] [steps] [sync_every]    (uses the oracle's constants: tools/ may import oracle/ like tests/)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import torch  # noqa: E402
import simq  # noqa: E402
import simq.learner as sl  # noqa: E402
from simq import synth  # noqa: E402

GAMMA, LR, MOMENTUM, WD, CLIP = 0.75, 0.01, 0.9, 1e-4, 100.0
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
sync_every = int(sys.argv[4]) if len(sys.argv) > 4 else 25
cin, cout = 5, 2
n0 = max(64, 2 * B)
trs = synth.make_transitions(n0 + steps, cin, cout, 11, terminal_frac=0.15)


def loop(fast):
    opts = {'deterministic': 1} if fast else {'deterministic': 1, 'fwd_overlap': 0, 'wgrad_overlap': 0}
    torch.manual_seed(3)
    policy = simq.FCN(cin, cout, precision=precision, options=opts)
    target = simq.FCN(cin, cout, precision=precision, options=opts)
    target.copy_state_from(policy)
    policy.train(); target.eval()
    ring = simq.DeviceReplayBuffer(4 * n0, cin, upload_stream=fast)
    for t in trs[:n0]:
        ring.push(*t)
    random.seed(5)
    so = sl.StepOptions(early_target_forward=fast, overlap_target_forward=fast)
    out = []
    for s in range(steps):
        ring.push(*trs[n0 + s])
        batch = ring.gather(ring.sample_indices(B))
        info = sl.train_step(policy, target, batch, GAMMA, B, LR, MOMENTUM, WD, CLIP, use_double_dqn=True, options=so)
        out.append((info['loss'], info['td_error']))
        if s % sync_every == sync_every - 1:
            target.copy_state_from(policy)
    torch.cuda.synchronize()
    return out, policy.flat_params.clone(), policy.bn_buffers.clone()


ref, p0, bn0 = loop(False)
got, p1, bn1 = loop(True)
first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a != b), None)
same = first is None and torch.equal(p0, p1) and torch.equal(bn0, bn1)
print('%s B=%d %d steps (target sync every %d): %s' % (precision, B, steps, sync_every,
      'bit-identical to the serial schedule (every loss and td error, final parameters and BatchNorm buffers)' if same else
      'DIFFERS: first loss mismatch at step %r (%r vs %r); max |dp| %.3e' % (first, ref[first] if first is not None else None,
                                                                            got[first] if first is not None else None, float((p0 - p1).abs().max()))))
sys.exit(0 if same else 1)
