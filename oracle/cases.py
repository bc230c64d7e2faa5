"""Shared case builders for the oracle fixtures (TEST INFRASTRUCTURE).

Both oracle/gen_golden.py (build container, reference mounted) and tests/ use
these so the inputs of every golden case are regenerated identically.
"""
import os
import sys
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), 'spatial-intention-maps_amd')
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from simq import synth  # noqa: E402  (numpy-only helper, no HIP)

from . import fcn, learner  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(_HERE), 'tests', 'golden')

# (name, cin, cout, batch, weight seed, data seed)
FORWARD_CASES = [('fwd_c4o2', 4, 2, 2, 11, 21), ('fwd_c5o2', 5, 2, 2, 12, 22), ('fwd_c5o1', 5, 1, 2, 13, 23)]
TRAIN_CASES = [('train_c4o2_b4', 4, 2, 4, 31, 41), ('train_c5o1_b4', 5, 1, 4, 32, 42), ('train_c4o2_b8', 4, 2, 8, 33, 43)]
# the reference's OTHER input-channel counts (tools_generate_experiments.py:200-204: num_input_channels in {3, 4, 5, 6, 7, 10}; 4 / 5 above):
# the stem's forward and weight gradient are the kernels whose addressing depends on Cin (K = 49 * Cin, unaligned rows; Cin = 10 leaves the
# dedicated fp32 stem kernel for the generic implicit GEMM).  Summaries only (no Q-map): < 100 KB per fixture.
TRAIN_CASES_CIN = [('train_c3o2_b4', 3, 2, 4, 91, 101), ('train_c6o2_b4', 6, 2, 4, 92, 102), ('train_c7o2_b4', 7, 2, 4, 93, 103),
                   ('train_c10o1_b4', 10, 1, 4, 94, 104)]
# the bench workload's own size (BASELINE configs[1]); summaries only, checked on the GPU without re-running the oracle
TRAIN_CASES_FULL = [('train_c4o2_b32', 4, 2, 32, 36, 46)]
# BASELINE configs[2] / configs[4]'s per-GPU shape (Cin 5, Cout 2, 128 transitions) and configs[3]'s two per-GPU shapes (Cin 5; lifting
# Cout 2 and pushing Cout 1, 64 transitions per GPU and net): reference train.train at the batch sizes the large-batch kernels are
# selected at, summaries + fp64 yardstick + the reference's own bf16-autocast error AT THAT SIZE (gen_golden.gen_train_sized)
TRAIN_CASES_SIZED = [('train_c5o2_b128', 5, 2, 128, 61, 71), ('train_c5o2_b64', 5, 2, 64, 62, 72), ('train_c5o1_b64', 5, 1, 64, 63, 73)]
# dense-upstream-gradient backward at those sizes (gen_golden.gen_dense_grad): loss = sum(Q * R) for a seeded dense R -- train-mode
# BatchNorm backward of a DENSE gradient does not cancel the way the TD loss's one-hot gradient does, so this is the well-conditioned
# network-level gradient case (fp32 ~1e-5, bf16 ~1e-2 against fp64) -- (name, cin, cout, batch, weight seed, data seed)
DENSE_GRAD_CASES = [('dense_c5o2_b128', 5, 2, 128, 64, 74), ('dense_c5o1_b64', 5, 1, 64, 65, 75)]


def dense_upstream(cout, batch, seed):
    """The dense dLoss/dQ of a DENSE_GRAD case: [B, Cout, 96, 96] float32, N(0,1) / sqrt(B * 96 * 96)."""
    rs = np.random.RandomState(seed + 15485863)
    return (rs.standard_normal((batch, cout, 96, 96)) / np.sqrt(batch * 96.0 * 96.0)).astype(np.float32)


# data-parallel emulation (SURVEY 8e / fixture G7): (name, cin, cout, global batch, shards, weight seed, data seed)
DP_CASES = [('dp_c5o2_b8_w1', 5, 2, 8, 1, 37, 50), ('dp_c5o2_b8_w2', 5, 2, 8, 2, 37, 50), ('dp_c5o2_b8_w4', 5, 2, 8, 4, 37, 50),
            ('dp_c5o2_b8_w8', 5, 2, 8, 8, 37, 50), ('dp_c5o1_b8_w2', 5, 1, 8, 2, 38, 48)]
# gradient-parity study (err_build <= k * err_reference-fp32): many seeded batches, judged as a distribution
GRAD_STUDY_CASES = [('gs_b8_%02d' % i, (4, 5)[i % 2], (2, 1)[(i // 2) % 2], 8, 200 + i, 300 + i) for i in range(10)] + \
                   [('gs_b32_%02d' % i, (4, 5, 4)[i], (2, 2, 1)[i], 32, 250 + i, 350 + i) for i in range(3)]
# the same study at configs[3]'s per-GPU batch (64): is the HIP fp32 gradient as accurate as the reference's at the larger sizes too?
GRAD_STUDY_B64_CASES = [('gs_b64_%02d' % i, 5, (2, 1)[i % 2], 64, 270 + i, 370 + i) for i in range(6)]
# ... and at configs[1]'s own batch (32, Cin 4, Cout 2: the headline workload): twelve more batches, for the decision which layers' grad-mode
# forward may run in F(4x4,3x3) (docs/history.md 4)
GRAD_STUDY_B32_CASES = [('gs_b32x_%02d' % i, 4, 2, 32, 400 + i, 500 + i) for i in range(12)]
# ... and at configs[2] / configs[4]'s per-GPU batch (128, Cin 5, Cout 2): twelve batches -- the single fixture train_c5o2_b128 sat at 2.6 x the
# reference's own error (one unlucky batch or a regression?); the distribution answers it (round 4)
GRAD_STUDY_B128_CASES = [('gs_b128_%02d' % i, 5, 2, 128, 600 + i, 700 + i) for i in range(12)]
INTENTION_CASES = [('intent_c5_b4', 5, 4, 34, 44), ('intent_c4_b3', 4, 3, 35, 45)]   # (name, cfg.num_input_channels, B, wseed, dseed)
SAMPLER_CASES = [(64, 4, 5), (10000, 32, 6), (10000, 1024, 7), (21, 21, 8)]

# the same with DataParallel's literal scatter of the COMPACTED next-state tensor (learner.dp_emulation_literal; gen_golden.gen_dp_literal)
DP_LITERAL_CASES = [('dplit_c5o2_b8_w2', 5, 2, 8, 2, 37, 50), ('dplit_c5o2_b8_w4', 5, 2, 8, 4, 37, 50), ('dplit_c5o1_b8_w2', 5, 1, 8, 2, 38, 48)]

LR, MOMENTUM, WEIGHT_DECAY, CLIP, GAMMA = 0.01, 0.9, 1e-4, 100, 0.75   # base config yml / train.py:186


def make_cfg(batch_size, use_double_dqn=True, grad_norm_clipping=CLIP):
    return SimpleNamespace(batch_size=batch_size, use_double_dqn=use_double_dqn,
                           grad_norm_clipping=grad_norm_clipping)


def oracle_state(cin, cout, seed, dtype=torch.float32):
    return fcn.state_from_numpy(synth.make_state_dict(cin, cout, seed), dtype)


def make_batch(cin, cout, batch, seed):
    trs = synth.make_transitions(batch, cin, cout, seed, terminal_frac=0.25)
    return learner.Transition(*zip(*trs))


def bn_buffer_vector(state):
    return np.concatenate([state[k].detach().double().numpy().ravel() for k in state
                           if k.endswith('running_mean') or k.endswith('running_var')])


def sample_indices(numel, n=16):
    """Deterministic element indices probed inside each gradient tensor."""
    return [int((i * 2654435761) % numel) for i in range(n)]


def grad_summary(grads):
    """Per-tensor L2 norm + 16 sampled elements (float64 arrays keyed by tensor name)."""
    out = OrderedDict()
    for k, g in grads.items():
        flat = g.detach().double().reshape(-1)
        idx = torch.tensor(sample_indices(flat.numel()))
        out[k] = np.concatenate([[float(flat.norm())], flat[idx].numpy()])
    return out


def param_summary(state, spec):
    """Per-tensor (sum, L2) of every parameter after the optimiser step."""
    rows = []
    for k, _, kind in spec:
        if fcn.is_parameter(kind):
            t = state[k].detach().double()
            rows.append([float(t.sum()), float(t.norm())])
    return np.asarray(rows)


def tracker_script(seed=5, steps=40, groups=(2, 1)):
    """Scripted collector episode for the TransitionTracker fixture: per step, which robots get an action, which ones
    receive a new observation, rewards, and episode ends.  Observations are small arrays tagged with a serial number."""
    rng = np.random.RandomState(seed)
    serial = [0]

    def obs():
        serial[0] += 1
        return np.full((2, 2, 1), float(serial[0]), dtype=np.float32)

    initial = [[obs() for _ in range(n)] for n in groups]
    script = []
    for t in range(steps):
        action = [[int(rng.randint(100)) if rng.rand() < 0.7 else None for _ in range(n)] for n in groups]
        done = bool(rng.rand() < 0.1)
        state = [[obs() if rng.rand() < 0.6 else None for _ in range(n)] for n in groups]
        reward = [[float(rng.randn()) for _ in range(n)] for n in groups]
        script.append((action, reward, state, done))
    return initial, script


def run_tracker(tracker_cls, seed=5):
    """Drive a TransitionTracker implementation through the script; returns per-buffer rows
    [state tag, action (-1 = None), reward, next_state tag (0 = None)]."""
    initial, script = tracker_script(seed)
    tr = tracker_cls(initial)
    rows = [[] for _ in initial]
    for action, reward, state, done in script:
        tr.update_action(action)
        for i, lst in enumerate(tr.update_step_completed(reward, state, done)):
            for (s, a, r, ns) in lst:
                rows[i].append([float(s[0, 0, 0]), -1.0 if a is None else float(a), float(r), 0.0 if ns is None else float(ns[0, 0, 0])])
    return [np.asarray(r, dtype=np.float64).reshape(-1, 4) for r in rows]


CKPT_CAPACITY, CKPT_CIN, CKPT_SEED = 3, 4, 77


def checkpoint_transitions():
    """Transitions for the checkpoint fixture (tests/golden/ref_checkpoint.pth.tar, pickled by the reference's own
    train.ReplayBuffer): 4 pushes into a capacity-3 ring (wraps by one), next_state of one transition is the state
    OBJECT of the following one as the collector produces them (train.py:61-66), one terminal transition."""
    from simq import synth
    o = list(synth.make_states(5, CKPT_CIN, CKPT_SEED))
    return [(o[0], 11, 0.5, o[1]), (o[1], 9216 + 7, -0.25, o[2]), (o[2], 123, 1.0, None), (o[3], 18431, 0.0, o[4])]
