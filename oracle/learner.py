"""Oracle: CPU restatement of the reference DQN learner.  TEST INFRASTRUCTURE.

Follows
  /root/reference/train.py:26       Transition
  /root/reference/train.py:28-45    ReplayBuffer (python-list ring + random.sample)
  /root/reference/train.py:108-141  train()  (double-DQN TD step, Huber, clip, SGD)
  torch.nn.utils.clip_grad_norm_    as called at train.py:134
  torch.optim.SGD(lr, momentum=0.9, weight_decay) as built at train.py:186

The optimiser and the clip are restated explicitly (same ATen op sequence as
torch's implementations) so the oracle documents the arithmetic the HIP
kernels must reproduce.
"""
import random
from collections import namedtuple, OrderedDict

import torch
from torch.nn.functional import smooth_l1_loss

from . import fcn

Transition = namedtuple('Transition', ('state', 'action', 'reward', 'next_state'))  # train.py:26


class ReplayBuffer:
    """train.py:28-45, verbatim semantics: ring over a python list, uniform
    sampling without replacement from the GLOBAL ``random`` stream."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.buffer = []
        self.position = 0

    def push(self, *args):
        if len(self.buffer) < self.capacity:
            self.buffer.append(None)
        self.buffer[self.position] = Transition(*args)
        self.position = (self.position + 1) % self.capacity

    def sample(self, batch_size):
        transitions = random.sample(self.buffer, batch_size)
        return Transition(*zip(*transitions))

    def __len__(self):
        return len(self.buffer)


def apply_transform(s):
    """policies.py:44-45 with torchvision ToTensor on a float32 HWC ndarray:
    HWC -> CHW view, no scaling (ToTensor only scales uint8), unsqueeze(0)."""
    return torch.from_numpy(s.transpose(2, 0, 1)).unsqueeze(0)


def grad_keys(state_spec):
    return [k for k, _, kind in state_spec if fcn.has_gradient(kind)]


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (norm_type 2): global L2
    norm of per-tensor L2 norms, coefficient max_norm/(total+1e-6) clamped to 1."""
    norms = [torch.linalg.vector_norm(g, 2.0) for g in grads]
    total_norm = torch.linalg.vector_norm(torch.stack(norms), 2.0)
    clip_coef = max_norm / (total_norm + 1e-6)
    clip_coef_clamped = torch.clamp(clip_coef, max=1.0)
    for g in grads:
        g.mul_(clip_coef_clamped)
    return total_norm


def sgd_step(params, grads, momentum_bufs, lr, momentum, weight_decay):
    """torch.optim.SGD.step() for one param group (dampening 0, no nesterov):
    g += wd*p ; buf = g (first step) | momentum*buf + g ; p -= lr*buf."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if weight_decay != 0:
            g = g.add(p, alpha=weight_decay)
        if momentum != 0:
            buf = momentum_bufs[i]
            if buf is None:
                buf = torch.clone(g).detach()
                momentum_bufs[i] = buf
            else:
                buf.mul_(momentum).add_(g, alpha=1)
            g = buf
        p.add_(g, alpha=-lr)


def td_targets(state, target_state, non_final_next_states, non_final_mask, reward_batch,
               batch_size, discount_factor, use_double_dqn):
    """train.py:116-126.  NOTE train.py:121: the double-DQN argmax forward runs the
    POLICY net in train mode under no_grad -> batch statistics + a second running-stat
    update; the target net runs in eval mode (train.py:216)."""
    dtype = reward_batch.dtype
    next_state_values = torch.zeros(batch_size, dtype=dtype)
    with torch.no_grad():
        n = non_final_next_states.size(0)
        if use_double_dqn:
            best_action = fcn.fcn_forward(state, non_final_next_states, True).view(n, -1).max(1)[1].view(n, 1)
            next_state_values[non_final_mask] = fcn.fcn_forward(
                target_state, non_final_next_states, False).view(n, -1).gather(1, best_action).view(-1)
        else:
            next_state_values[non_final_mask] = fcn.fcn_forward(
                target_state, non_final_next_states, False).view(n, -1).max(1)[0]
    return reward_batch + discount_factor * next_state_values


def train_step(cfg, state, target_state, spec, momentum_bufs, batch, discount_factor,
               lr, momentum, weight_decay, dtype=torch.float32, extras=None):
    """One reference ``train()`` call (train.py:108-141) on functional state.

    ``state`` / ``target_state``: dicts keyed as fcn.state_spec (mutated in place:
    parameters by SGD, BN buffers by the two train-mode forwards).
    ``momentum_bufs``: list aligned with grad_keys(spec) (None before the first step).
    ``extras``: optional dict receiving grads (pre-clip), total_norm, q, y.
    Returns {'td_error': float, 'loss': float}.
    """
    B = cfg.batch_size
    state_batch = torch.cat([apply_transform(s) for s in batch.state]).to(dtype)
    action_batch = torch.tensor(batch.action, dtype=torch.long)
    reward_batch = torch.tensor(batch.reward, dtype=torch.float32).to(dtype)
    non_final_next_states = torch.cat([apply_transform(s) for s in batch.next_state if s is not None]).to(dtype)
    non_final_mask = torch.tensor(tuple(map(lambda s: s is not None, batch.next_state)), dtype=torch.bool)

    gkeys = grad_keys(spec)
    params = [state[k] for k in gkeys]
    for p in params:
        p.requires_grad_(True)
        p.grad = None
    try:
        output = fcn.fcn_forward(state, state_batch, True)                       # train.py:114
        q = output.view(B, -1).gather(1, action_batch.unsqueeze(1)).squeeze(1)   # train.py:115
        y = td_targets(state, target_state, non_final_next_states, non_final_mask, reward_batch,
                       B, discount_factor, cfg.use_double_dqn)                    # train.py:116-126
        td_error = torch.abs(q - y).detach()                                      # train.py:127
        loss = smooth_l1_loss(q, y)                                               # train.py:129
        grads = list(torch.autograd.grad(loss, params))                           # train.py:131-132
    finally:
        for p in params:
            p.requires_grad_(False)
    if extras is not None:
        extras['grads'] = OrderedDict((k, g.clone()) for k, g in zip(gkeys, grads))
        extras['q'] = q.detach().clone()
        extras['y'] = y.detach().clone()
        extras['output'] = output.detach()
    total_norm = None
    if cfg.grad_norm_clipping is not None:                                        # train.py:133-134
        total_norm = clip_grad_norm(grads, cfg.grad_norm_clipping)
    if extras is not None:
        extras['total_norm'] = None if total_norm is None else float(total_norm)
    with torch.no_grad():
        sgd_step(params, grads, momentum_bufs, lr, momentum, weight_decay)        # train.py:135
    return {'td_error': td_error.mean().item(), 'loss': loss.item()}              # train.py:137-139


def train_intention_step(state, spec, momentum_bufs, batch, lr, momentum, weight_decay, dtype=torch.float32,
                         extras=None):
    """One reference ``train_intention()`` call (train.py:143-158) on functional state.

    ``state``: intention-net dict, FCN(num_input_channels - 1, 1) (policies.py:91-95); the LAST channel of every
    replay state is the ground-truth intention map (train.py:144).  No gradient clipping on this path.
    Returns {'loss_intention': float}.
    """
    state_batch = torch.cat([apply_transform(s[:, :, :-1]) for s in batch.state]).to(dtype)      # train.py:145
    target_batch = torch.cat([apply_transform(s[:, :, -1:]) for s in batch.state]).to(dtype)     # train.py:146
    gkeys = grad_keys(spec)
    params = [state[k] for k in gkeys]
    for p in params:
        p.requires_grad_(True)
        p.grad = None
    try:
        output = fcn.fcn_forward(state, state_batch, True)                                        # train.py:148
        loss = torch.nn.functional.binary_cross_entropy_with_logits(output, target_batch)         # train.py:149-150
        grads = list(torch.autograd.grad(loss, params))                                           # train.py:151-152
    finally:
        for p in params:
            p.requires_grad_(False)
    if extras is not None:
        extras['grads'] = OrderedDict((k, g.clone()) for k, g in zip(gkeys, grads))
        extras['output'] = output.detach()
    with torch.no_grad():
        sgd_step(params, grads, momentum_bufs, lr, momentum, weight_decay)                        # train.py:153
    return {'loss_intention': loss.item()}                                                        # train.py:155-156


def forward_backward(state, spec, state_batch, action_batch, y, dtype=torch.float32):
    """M1 of SURVEY 8d: policy forward (train-mode BN) + gather + Huber + backward,
    no target computation / optimiser.  Used for the cpu_baseline timing and M1 parity."""
    B = state_batch.shape[0]
    gkeys = grad_keys(spec)
    params = [state[k] for k in gkeys]
    for p in params:
        p.requires_grad_(True)
    try:
        output = fcn.fcn_forward(state, state_batch.to(dtype), True)
        q = output.view(B, -1).gather(1, action_batch.unsqueeze(1)).squeeze(1)
        loss = smooth_l1_loss(q, y.to(dtype))
        grads = torch.autograd.grad(loss, params)
    finally:
        for p in params:
            p.requires_grad_(False)
    return loss.item(), q.detach(), OrderedDict(zip(gkeys, grads)), output.detach()


# ---- data-parallel emulation (fixture G7) ------------------------------------------------
def shard_gradients(cfg, state, target_state, spec, shard_batch, global_batch, discount_factor, update_buffers,
                    dtype=torch.float32):
    """What ONE DataParallel replica contributes (policies.py:39): forward of its shard with its own
    train-mode BN statistics, Huber SUM over the shard divided by the GLOBAL batch, backward.
    Returns (flat gradient in grad_keys order, [sum huber, sum |td|])."""
    B = len(shard_batch.action)
    state_batch = torch.cat([apply_transform(s) for s in shard_batch.state]).to(dtype)
    action_batch = torch.tensor(shard_batch.action, dtype=torch.long)
    reward_batch = torch.tensor(shard_batch.reward, dtype=torch.float32).to(dtype)
    nf = [apply_transform(s) for s in shard_batch.next_state if s is not None]
    mask = torch.tensor(tuple(s is not None for s in shard_batch.next_state), dtype=torch.bool)
    gkeys = grad_keys(spec)
    params = [state[k] for k in gkeys]
    for p in params:
        p.requires_grad_(True)
    try:
        output = fcn.fcn_forward(state, state_batch, True, update_buffers=update_buffers)
        q = output.view(B, -1).gather(1, action_batch.unsqueeze(1)).squeeze(1)
        nsv = torch.zeros(B, dtype=dtype)
        if nf:
            nfns = torch.cat(nf).to(dtype)
            n = nfns.size(0)
            with torch.no_grad():
                best = fcn.fcn_forward(state, nfns, True, update_buffers=update_buffers).view(n, -1).max(1)[1].view(n, 1)
                nsv[mask] = fcn.fcn_forward(target_state, nfns, False).view(n, -1).gather(1, best).view(-1)
        y = reward_batch + discount_factor * nsv
        huber = smooth_l1_loss(q, y, reduction='sum')
        grads = torch.autograd.grad(huber / global_batch, params)
    finally:
        for p in params:
            p.requires_grad_(False)
    flat = torch.cat([g.reshape(-1) for g in grads])
    return flat, torch.stack([huber.detach(), torch.abs(q - y).detach().sum()])


def dp_emulation(cfg, state, target_state, spec, batch, world_size, discount_factor, dtype=torch.float32):
    """Single-process emulation of an N-replica step: contiguous shards (torch.chunk sizes), per-shard BN
    statistics, gradient sum, rank 0's running statistics kept.  Returns (flat grad sum, loss, td_error)."""
    gB = len(batch.action)
    chunk = -(-gB // world_size)
    total, sums = None, torch.zeros(2, dtype=dtype)
    for r in range(world_size):
        lo, hi = min(r * chunk, gB), min((r + 1) * chunk, gB)
        if lo == hi:
            continue
        shard = Transition(*[f[lo:hi] for f in batch])
        flat, s = shard_gradients(cfg, state, target_state, spec, shard, gB, discount_factor, update_buffers=(r == 0), dtype=dtype)
        total = flat if total is None else total + flat
        sums = sums + s
    return total, float(sums[0]) / gB, float(sums[1]) / gB


def dp_emulation_literal(cfg, state, target_state, spec, batch, world_size, discount_factor, dtype=torch.float32):
    """nn.DataParallel LITERALLY (policies.py:39 around train.py:114-132): like dp_emulation, except for the double-DQN forward
    `policy_net(non_final_next_states)` (train.py:121) -- DataParallel scatters the tensor IT IS GIVEN, the COMPACTED non-final next
    states, in torch.chunk pieces of ceil(N'/world) rows, so replica r picks the greedy actions of compacted chunk r (with ITS OWN
    train-mode BatchNorm statistics over that chunk; replica 0's running statistics see chunk 0), whichever transitions of the
    minibatch those rows came from.  The target net runs in eval mode: its chunking does not matter.
    Returns (flat grad sum, loss, td_error, best actions [N'])."""
    gB = len(batch.action)
    chunk = -(-gB // world_size)
    gkeys = grad_keys(spec)
    params = [state[k] for k in gkeys]
    action_batch = torch.tensor(batch.action, dtype=torch.long)
    reward_batch = torch.tensor(batch.reward, dtype=torch.float32).to(dtype)
    nf = [apply_transform(s) for s in batch.next_state if s is not None]
    mask = torch.tensor(tuple(s is not None for s in batch.next_state), dtype=torch.bool)
    for p in params:
        p.requires_grad_(True)
    try:
        # train.py:114 -- replica r forwards rows [r * chunk, (r + 1) * chunk) of the minibatch
        outs = []
        for r in range(world_size):
            lo, hi = min(r * chunk, gB), min((r + 1) * chunk, gB)
            if lo == hi:
                continue
            sb = torch.cat([apply_transform(s) for s in batch.state[lo:hi]]).to(dtype)
            outs.append(fcn.fcn_forward(state, sb, True, update_buffers=(r == 0)).view(hi - lo, -1))
        q = torch.cat(outs).gather(1, action_batch.unsqueeze(1)).squeeze(1)                        # train.py:115
        nsv = torch.zeros(gB, dtype=dtype)
        best = torch.zeros(0, dtype=torch.long)
        if nf:
            nfns = torch.cat(nf).to(dtype)
            n = nfns.size(0)
            nchunk = -(-n // world_size)
            with torch.no_grad():
                bests = []
                for r in range(world_size):
                    lo, hi = min(r * nchunk, n), min((r + 1) * nchunk, n)
                    if lo == hi:
                        continue
                    bests.append(fcn.fcn_forward(state, nfns[lo:hi], True, update_buffers=(r == 0)).view(hi - lo, -1).max(1)[1])
                best = torch.cat(bests)
                nsv[mask] = fcn.fcn_forward(target_state, nfns, False).view(n, -1).gather(1, best.view(n, 1)).view(-1)
        y = reward_batch + discount_factor * nsv
        huber = smooth_l1_loss(q, y, reduction='sum')
        # DataParallel's backward: every replica differentiates ITS rows (upstream gradient huber'(q - y) / gB), the replicas' gradients
        # are reduce-added onto device 0 in replica order
        flat, lo = None, 0
        for o in outs:
            hi = lo + o.shape[0]
            q_r = o.gather(1, action_batch[lo:hi].unsqueeze(1)).squeeze(1)
            g_r = torch.autograd.grad(smooth_l1_loss(q_r, y[lo:hi], reduction='sum') / gB, params)
            f_r = torch.cat([g.reshape(-1) for g in g_r])
            flat = f_r if flat is None else flat + f_r
            lo = hi
    finally:
        for p in params:
            p.requires_grad_(False)
    return flat, float(huber.detach()) / gB, float(torch.abs(q - y).detach().sum()) / gB, best, q.detach(), y.detach()
