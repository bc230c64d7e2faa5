"""CPU: the multi-process collector's pipe protocol (train_multiprocess.py:147-275) with a stand-in policy -- no GPU involved.
Worker processes are spawned, run simq.synth.SyntheticEnv and never import torch / libsimq."""
import types

import numpy as np
import pytest

from simq.collector import Collector, CollectWorker
from simq.synth import synthetic_env_from_cfg


class ScriptedPolicy:
    """policy.step / step_many stand-in: action = 7 for every robot that awaits one."""

    def __init__(self):
        self.calls, self.batched_calls = 0, 0

    def step(self, state, exploration_eps=None):
        self.calls += 1
        return [[None if s is None else 7 for s in g] for g in state]

    def step_many(self, states, exploration_eps=None):
        self.batched_calls += 1
        return [[[None if s is None else 7 for s in g] for g in st] for st in states]


def failing_env(cfg, worker_index):
    raise RuntimeError('simulator failed to start in worker %d' % worker_index)


def make_cfg():
    return types.SimpleNamespace(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}], num_input_channels=4, seed=11, episode_len=5)


def same(t1, t2):
    return all(len(a) == len(b) and all(np.array_equal(x[0], y[0]) and x[1] == y[1] and x[2] == y[2] and
                                        ((x[3] is None and y[3] is None) or np.array_equal(x[3], y[3])) for x, y in zip(a, b))
               for a, b in zip(t1, t2))


def test_round_robin_and_batched_service_match_the_in_process_worker():
    cfg, pol = make_cfg(), ScriptedPolicy()
    col = Collector(cfg, pol, num_workers=2, env_fn=synthetic_env_from_cfg)
    try:
        # reference workers for both environments, stepped in-process with the same scripted actions
        refs = [CollectWorker(cfg, synthetic_env_from_cfg, w) for w in range(2)]
        first = [col.step(0.0) for _ in range(2)]
        assert first == [([], False), ([], False)]               # the workers' hello messages (train_multiprocess.py:171)
        for k in range(12):                                       # call k returns the env step the worker did for its previous action
            w = k % 2
            tr, done = col.step(0.0)
            want_tr, want_done, _ = refs[w].step(pol.step(refs[w].get_state()))
            assert done == want_done and same(tr, want_tr)
        for _ in range(5):                                        # batched service: both workers per call
            res = col.step_all(0.0)
            assert len(res) == 2
            for w, (tr, done) in enumerate(res):
                want_tr, want_done, _ = refs[w].step(pol.step(refs[w].get_state()))
                assert done == want_done and same(tr, want_tr)
        assert pol.batched_calls == 5
    finally:
        col.close()
    assert all(not w.is_alive() for w in col.workers)


def test_worker_failures_surface_in_the_parent():
    col = Collector(make_cfg(), ScriptedPolicy(), num_workers=1, env_fn=failing_env)
    with pytest.raises(RuntimeError, match='simulator failed to start in worker 0'):
        col.step(0.0)
    for w in col.workers:
        w.join(timeout=10)


def test_env_fn_is_required():
    with pytest.raises(ValueError, match='env_fn'):
        Collector(make_cfg(), ScriptedPolicy(), num_workers=None)
