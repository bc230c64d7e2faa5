#!/usr/bin/env python3
"""GPU box: a long training loop shaped like train.py:241-269 (push one transition, sample from the HBM ring, train(), copy the policy's weights
into the target net every `sync_every` steps) run TWICE on deterministic plans -- once with the bench's schedule (three forwards side by side,
piped weight gradients, upload stream, early target forward held to block 4, the loss read without waiting for the backward pass) and once
fully serial (fwd_overlap = wgrad_overlap = 0, no upload stream, no early forward) -- and compared BIT FOR BIT after every step.  The suite's
tests/test_gpu_overlap.py does this for 2-6 steps; a buffer-reuse or ordering race that needs an unlucky interleaving shows up over hundreds.
usage: soak_schedule.py [fp32|bf16] [batch] [steps] [sync_every] [single|dp1|groups]
  single  one learner (default)
  dp1     the fast loop runs the DATA-PARALLEL form of the step on a 1-rank RCCL communicator (backward phases, gradient buckets and the loss sums
          all-reduced, loss copied from the communicator's stream) -- against the plain serial step: the same arithmetic in the same order
  groups  two robot groups (Cout 2 and 1), each on a launch stream of its own with every step enqueued before the first loss is read
          (simq.train_groups(concurrent=True)'s mechanism) -- against the two groups one after the other, serial schedule"""
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
mode = sys.argv[5] if len(sys.argv) > 5 else 'single'
cin = 5
couts = (2, 1) if mode == 'groups' else (2,)
comm = None
if mode == 'dp1':
    import torch.distributed as dist
    from simq import dist as sdist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29553')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    comm = sdist.Comm(dist.group.WORLD)
n0 = max(64, 2 * B)
trs = [synth.make_transitions(n0 + steps, cin, co, 11 + gi, terminal_frac=0.15) for gi, co in enumerate(couts)]


def loop(fast):
    opts = {'deterministic': 1} if fast else {'deterministic': 1, 'fwd_overlap': 0, 'wgrad_overlap': 0}
    torch.manual_seed(3)
    groups = []
    for gi, co in enumerate(couts):
        policy = simq.FCN(cin, co, precision=precision, options=opts)
        target = simq.FCN(cin, co, precision=precision, options=opts)
        target.copy_state_from(policy)
        policy.train(); target.eval()
        ring = simq.DeviceReplayBuffer(4 * n0, cin, upload_stream=fast)
        for t in trs[gi][:n0]:
            ring.push(*t)
        if mode == 'groups':
            sl.learner_streams(policy, own_launch_stream=fast)
        groups.append((policy, target, ring))
    random.seed(5)
    so = sl.StepOptions(early_target_forward=fast, overlap_target_forward=fast)
    out = []
    for s in range(steps):
        pend = []
        for gi, (policy, target, ring) in enumerate(groups):
            launch = sl.learner_streams(policy).launch
            with torch.cuda.stream(launch if launch is not None else torch.cuda.current_stream()):
                ring.push(*trs[gi][n0 + s])
                batch = ring.gather(ring.sample_indices(B))
                pend.append(sl.train_step(policy, target, batch, GAMMA, B, LR, MOMENTUM, WD, CLIP, use_double_dqn=True, options=so,
                                          sync='defer' if (fast and mode == 'groups') else True,
                                          **(dict(global_batch=B, comm=comm) if (fast and comm is not None) else {})))
        for info in pend:
            info = info.result() if hasattr(info, 'result') else info
            out.append((info['loss'], info['td_error']))
        if s % sync_every == sync_every - 1:
            for policy, target, _ in groups:
                target.copy_state_from(policy)
    torch.cuda.synchronize()
    return out, torch.cat([g[0].flat_params for g in groups]).clone(), torch.cat([g[0].bn_buffers for g in groups]).clone()


ref, p0, bn0 = loop(False)
got, p1, bn1 = loop(True)
first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a != b), None)
same = first is None and torch.equal(p0, p1) and torch.equal(bn0, bn1)
print('%s %s B=%d %d steps (target sync every %d): %s' % (mode, precision, B, steps, sync_every,
      'bit-identical to the serial schedule (every loss and td error, final parameters and BatchNorm buffers)' if same else
      'DIFFERS: first loss mismatch at step %r (%r vs %r); max |dp| %.3e' % (first, ref[first] if first is not None else None,
                                                                            got[first] if first is not None else None, float((p0 - p1).abs().max()))))
if comm is not None:
    comm.close()
    dist.destroy_process_group()
sys.exit(0 if same else 1)
