"""Experience collection across processes: drop-in for CollectWorker / Collector of train_multiprocess.py:147-275.

The environments (CPU: simulator stepping, observation rendering) run in worker processes; the learner process owns the GPU
and serves them.  Wire protocol per worker, as in the reference: the worker sends (transitions_per_buffer, done, logging_info)
of its last step, then its current state, and blocks until it receives the next action (or 'close').

  Collector.step(eps)      the reference's round-robin: serve ONE worker per call with policy.step (train_multiprocess.py:254-264)
  Collector.step_all(eps)  MI355X form: serve EVERY worker in one call -- all awaiting robots of all
                           environments share one eval forward + argmax per robot group (DQNPolicy.step_many), so the GPU sees
                           one batched launch sequence instead of num_workers batch-1 ones, and all environments step concurrently
                           while the learner trains.

The simulator itself is out of scope: `env_fn(cfg, worker_index)` builds the environment inside the worker
(utils.get_env_from_cfg in the reference; simq.synth.synthetic_env_from_cfg for the tests).  Workers never touch the GPU and
are started with the 'spawn' method, so the parent's HIP context is not inherited.  Transitions cross the process boundary by
pickle: the ndarray that is `next_state` of one message and `state` of a later one arrives as two objects, so feed them to a
DeviceReplayBuffer (or an AliasedDeviceReplayBuffer with pool_slots = 2 * capacity).
"""
import multiprocessing as mp
import time
import traceback

from .tracker import TransitionTracker


class CollectWorker:
    """train_multiprocess.py:147-208.  In-process use: CollectWorker(cfg, env_fn).  As a process: Collector starts `_worker_main`."""

    def __init__(self, cfg, env_fn, worker_index=0):
        self.cfg, self.worker_index = cfg, worker_index
        self.env = env_fn(cfg, worker_index)
        self.state = self.env.reset()
        self.transition_tracker = TransitionTracker(self.state)

    def get_state(self):
        return self.state

    def step(self, action):
        self.transition_tracker.update_action(action)
        self.state, reward, done, info = self.env.step(action)
        transitions_per_buffer = self.transition_tracker.update_step_completed(reward, self.state, done)
        logging_info = None
        if done:
            logging_info = {'scalars': {'total/%s' % k: v for k, v in info.items() if isinstance(v, (int, float))}, 'images': {}}
            self.state = self.env.reset()
            self.transition_tracker = TransitionTracker(self.state)
        return transitions_per_buffer, done, logging_info

    def close(self):
        close = getattr(self.env, 'close', None)
        if close is not None:
            close()


def _worker_main(cfg, env_fn, worker_index, conn):
    try:
        worker = CollectWorker(cfg, env_fn, worker_index)
        conn.send(([], False, None))                             # transitions_per_buffer, done, logging_info
        while True:
            conn.send(worker.get_state())
            action = conn.recv()
            if isinstance(action, str) and action == 'close':
                worker.close()
                break
            conn.send(worker.step(action))
    except Exception as e:                                       # reported to the parent, raised there (train_multiprocess.py:179-181)
        conn.send((e, traceback.format_exc()))


class Collector:
    """train_multiprocess.py:210-275.  `logger` may be None; otherwise it gets .scalar / .image / .update calls as in the reference."""

    def __init__(self, cfg, policy, logger=None, num_workers=None, env_fn=None):
        if env_fn is None:
            raise ValueError('Collector: env_fn(cfg, worker_index) is required (the simulator is not part of this package)')
        self.cfg, self.policy, self.logger, self.num_workers = cfg, policy, logger, num_workers
        if num_workers is not None:
            ctx = mp.get_context('spawn')
            self.curr_worker_index = 0
            self.workers, self.conns = [], []
            for i in range(num_workers):
                parent_conn, child_conn = ctx.Pipe()
                w = ctx.Process(target=_worker_main, args=(cfg, env_fn, i, child_conn), daemon=True)
                w.start()
                child_conn.close()
                self.workers.append(w)
                self.conns.append(parent_conn)
        else:
            self.worker = CollectWorker(cfg, env_fn)

    @staticmethod
    def _recv_result(conn):
        r = conn.recv()
        if isinstance(r[0], Exception):
            e, tb = r
            raise e from Exception(tb)
        return r

    def _log(self, done, logging_info, t0):
        if self.logger is None:
            return
        if done and logging_info:
            for name, val in logging_info['scalars'].items():
                self.logger.scalar(name, val)
            for name, val in logging_info['images'].items():
                self.logger.image(name, val)
        self.logger.update('timing/collect_time', time.time() - t0, add_hostname=True)

    def step(self, exploration_eps):
        """One worker per call -> (transitions_per_buffer, done)."""
        t0 = time.time()
        if self.num_workers is None:
            action = self.policy.step(self.worker.get_state(), exploration_eps=exploration_eps)
            transitions, done, info = self.worker.step(action)
        else:
            conn = self.conns[self.curr_worker_index]
            transitions, done, info = self._recv_result(conn)
            state = conn.recv()
            conn.send(self.policy.step(state, exploration_eps=exploration_eps))
            self.curr_worker_index = (self.curr_worker_index + 1) % self.num_workers
        self._log(done, info, t0)
        return transitions, done

    def step_all(self, exploration_eps):
        """Every worker in one call -> list of (transitions_per_buffer, done), one per worker, actions from ONE batched forward."""
        if self.num_workers is None:
            return [self.step(exploration_eps)]
        t0 = time.time()
        results = [self._recv_result(c) for c in self.conns]
        states = [c.recv() for c in self.conns]
        for c, a in zip(self.conns, self.policy.step_many(states, exploration_eps=exploration_eps)):
            c.send(a)
        for _, done, info in results:
            self._log(done, info, t0)
        return [(tr, done) for tr, done, _ in results]

    def close(self):
        if self.num_workers is None:
            self.worker.close()
            return
        for conn in self.conns:
            conn.recv()
            conn.recv()
            conn.send('close')
        for w in self.workers:
            w.join(timeout=30)
