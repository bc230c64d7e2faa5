"""Seeded synthetic weights / replay data (numpy only; no torch, no HIP).

Used by bench.py, the tests and oracle/gen_golden.py so that every side
regenerates identical inputs from ``numpy.random.RandomState(seed)`` and only
results need to be stored as fixtures.
"""
from collections import OrderedDict

import numpy as np

from . import arch


def make_state_dict(num_input_channels, num_output_channels, seed):
    """Reference-keyed numpy state dict with non-trivial BN statistics.

    Draw order == state_spec order (frozen stream).  Conv weights ~ N(0, sqrt(2/(Cout*k*k)))
    (kaiming fan_out, resnet.py:72), BN gamma 1+-0.1, beta +-0.1, running_mean N(0,0.1),
    running_var U(0.5,1.5), conv biases U(-0.05,0.05), fc N(0,0.01).
    """
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, kind in arch.state_spec(num_input_channels, num_output_channels):
        if kind == 'conv_w':
            std = np.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
            v = rs.standard_normal(shape) * std
        elif kind == 'conv_b':
            v = rs.uniform(-0.05, 0.05, shape)
        elif kind == 'bn_weight':
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif kind == 'bn_bias':
            v = 0.1 * rs.standard_normal(shape)
        elif kind == 'bn_mean':
            v = 0.1 * rs.standard_normal(shape)
        elif kind == 'bn_var':
            v = rs.uniform(0.5, 1.5, shape)
        elif kind == 'bn_count':
            sd[key] = np.array(0, dtype=np.int64)
            continue
        elif kind == 'fc_w':
            v = 0.01 * rs.standard_normal(shape)
        elif kind == 'fc_b':
            v = np.zeros(shape)
        else:
            raise ValueError(kind)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def make_states(n, num_input_channels, seed):
    """n synthetic overhead states, float32 HWC [96,96,C] (envs.py:2183 layout).

    Channel value distributions follow SURVEY 8d: ch0 overhead map in {0,1/8..1},
    ch1 robot map in {0,.5,1}, remaining channels smooth ramps in [0,0.6].
    """
    rs = np.random.RandomState(seed)
    W = arch.STATE_WIDTH
    out = np.empty((n, W, W, num_input_channels), dtype=np.float32)
    yy, xx = np.mgrid[0:W, 0:W].astype(np.float32) / (W - 1)
    for i in range(n):
        for c in range(num_input_channels):
            if c == 0:
                out[i, :, :, c] = rs.randint(0, 9, (W, W)).astype(np.float32) / 8.0
            elif c == 1:
                out[i, :, :, c] = rs.randint(0, 3, (W, W)).astype(np.float32) / 2.0
            else:
                a, b, ph = rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(0, 1)
                ramp = np.abs(a * yy + b * xx + ph)
                out[i, :, :, c] = (0.6 * ramp / max(ramp.max(), 1e-6)).astype(np.float32)
    return out


def make_transitions(n, num_input_channels, num_output_channels, seed, terminal_frac=0.1):
    """n synthetic transitions as reference-style tuples
    (state HWC f32, action int, reward float, next_state HWC f32 | None)."""
    rs = np.random.RandomState(seed + 7919)
    states = make_states(n, num_input_channels, seed)
    next_states = make_states(n, num_input_channels, seed + 104729)
    W = arch.STATE_WIDTH
    actions = rs.randint(0, num_output_channels * W * W, n)
    rewards = np.clip(rs.standard_normal(n), -1.5, 2.0).astype(np.float32)
    terminal = rs.uniform(size=n) < terminal_frac
    if n > 0 and terminal.all():
        terminal[0] = False      # train.py:112 needs >=1 non-final next state
    out = []
    for i in range(n):
        out.append((states[i], int(actions[i]), float(rewards[i]), None if terminal[i] else next_states[i]))
    return out


class SyntheticEnv:
    """reset() -> state ; step(action) -> (state, reward, done, info) with the nested [group][robot] lists of envs.py."""

    def __init__(self, robot_config, channels, seed, episode_len=25):
        self.groups = [next(iter(g.values())) for g in robot_config]
        self.C, self.rng, self.episode_len, self.t = channels, np.random.RandomState(seed), episode_len, 0

    def _obs(self):
        return self.rng.rand(arch.STATE_WIDTH, arch.STATE_WIDTH, self.C).astype(np.float32)

    def reset(self):
        self.t = 0
        return [[self._obs() for _ in range(n)] for n in self.groups]

    def step(self, action):
        self.t += 1
        done = self.t >= self.episode_len
        state = [[self._obs() if (self.rng.rand() < 0.7 and not done) else None for _ in range(n)] for n in self.groups]
        if not done and all(s is None for g in state for s in g):
            state[0][0] = self._obs()
        reward = [[float(np.clip(self.rng.randn(), -1.5, 2.0)) for _ in range(n)] for n in self.groups]
        return state, reward, done, {'steps': self.t}


def synthetic_env_from_cfg(cfg, worker_index=0):
    """Environment factory for simq.collector (stands where utils.get_env_from_cfg is in train_multiprocess.py:162)."""
    return SyntheticEnv(cfg.robot_config, cfg.num_input_channels, seed=getattr(cfg, 'seed', 0) + 1000 * worker_index,
                        episode_len=getattr(cfg, 'episode_len', 25))
