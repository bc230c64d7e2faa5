#!/usr/bin/env python3
"""The reference's training loop (train.py:main, :181-346) on the simq drop-ins, with a synthetic stand-in for the
pybullet environment (random 96x96xC observations, random rewards, robots that finish their actions at random times).

It exists to exercise the drop-in surface end to end on an MI355X: DQNPolicy / DQNIntentionPolicy.step, TransitionTracker,
(Aliased)DeviceReplayBuffer.push / sample, train_groups (train + train_intention per robot group, concurrently), target sync through state_dict, the Q-map debug path
of train.py:294-296, policy + optimizer checkpoints and resume.  Not a benchmark (bench.py is)."""
import argparse
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import simq  # noqa: E402
from simq import arch  # noqa: E402


from simq.synth import SyntheticEnv  # noqa: E402,F401


def run(cfg, checkpoint_dir, verbose=True):
    env = SyntheticEnv(cfg.robot_config, cfg.num_input_channels, seed=cfg.seed)
    num_robot_groups = len(cfg.robot_config)
    Policy = simq.DQNIntentionPolicy if cfg.use_predicted_intention else simq.DQNPolicy
    policy = Policy(cfg, train=True, random_seed=cfg.seed)                                            # train.py:181
    sgd = lambda net: torch.optim.SGD(net.parameters(), lr=cfg.learning_rate, momentum=0.9, weight_decay=cfg.weight_decay)
    optimizers = [sgd(n) for n in policy.policy_nets]                                                # train.py:184-186
    optimizers_intention = [sgd(n) for n in policy.intention_nets] if cfg.use_predicted_intention else None
    replay_buffers = [simq.AliasedDeviceReplayBuffer(cfg.replay_buffer_size, cfg.num_input_channels)
                      for _ in range(num_robot_groups)]                                               # train.py:193-195
    start_timestep = 0
    if cfg.checkpoint_path is not None:                                                               # train.py:200-211
        start_timestep, _, replay_buffers = simq.resume(cfg.checkpoint_path, optimizers, cfg.num_input_channels,
                                                        optimizers_intention=optimizers_intention)
    target_nets = policy.build_policy_nets()                                                          # train.py:213-216
    for i in range(num_robot_groups):
        target_nets[i].load_state_dict(policy.policy_nets[i].state_dict())
        target_nets[i].eval()
    state = env.reset()
    tracker = simq.TransitionTracker(state)
    learning_starts = int(round(cfg.learning_starts_frac * cfg.total_timesteps))
    total = learning_starts + cfg.total_timesteps
    log = []
    for timestep in range(start_timestep, total):
        eps = 1 - (1 - cfg.final_exploration) * min(1, max(0, timestep - learning_starts) / (cfg.exploration_frac * cfg.total_timesteps))
        if cfg.use_predicted_intention:                                                               # train.py:229-233
            gt = max(0, timestep - learning_starts) / cfg.total_timesteps <= cfg.use_predicted_intention_frac
            action = policy.step(state, exploration_eps=eps, use_ground_truth_intention=gt)
        else:
            action = policy.step(state, exploration_eps=eps)
        tracker.update_action(action)
        state, reward, done, info = env.step(action)
        for i, transitions in enumerate(tracker.update_step_completed(reward, state, done)):         # train.py:241-244
            for transition in transitions:
                replay_buffers[i].push(*transition)
        if done:
            state = env.reset()
            tracker = simq.TransitionTracker(state)
        if timestep >= learning_starts and (timestep + 1) % cfg.train_freq == 0:                      # train.py:252-264
            # the loop pass over the robot groups (train.py:255-261) handed over whole: every group's train() -- and train_intention() -- on a
            # launch stream of its own, side by side on the device (simq.train_groups; per net the results of the sequential loop)
            batches = [replay_buffers[i].sample(cfg.batch_size) if len(replay_buffers[i]) >= cfg.batch_size else None for i in range(num_robot_groups)]
            infos = simq.train_groups(cfg, policy.policy_nets, target_nets, optimizers, batches, policy.apply_transform, cfg.discount_factors,
                                      intention_nets=policy.intention_nets if cfg.use_predicted_intention else None,
                                      optimizers_intention=optimizers_intention, concurrent=True)
            for i, info_i in enumerate(infos):
                if info_i is None:
                    continue
                log.append((timestep + 1, i, info_i))
                if verbose:
                    print('t=%d group %d %s' % (timestep + 1, i, {k: round(v, 4) for k, v in info_i.items()}))
        if (timestep + 1) % cfg.target_update_freq == 0:                                              # train.py:267-269
            for i in range(num_robot_groups):
                target_nets[i].load_state_dict(policy.policy_nets[i].state_dict())
        if done and timestep >= learning_starts and all(len(b) for b in replay_buffers):             # train.py:292-296
            random_state = [[random.choice(replay_buffers[i].buffer).state] for i in range(num_robot_groups)]
            _, dbg = policy.step(random_state, debug=True)       # _DeviceObs handles convert on demand
            assert all(dbg['output'][i][0].shape[-2:] == (arch.STATE_WIDTH, arch.STATE_WIDTH) for i in range(num_robot_groups))
    policy_path = simq.save_policy(checkpoint_dir, total, policy.policy_nets,                         # train.py:312-345
                                   policy.intention_nets if cfg.use_predicted_intention else None)
    checkpoint_path = simq.save_checkpoint(checkpoint_dir, total, 0, optimizers, replay_buffers, optimizers_intention)
    return policy, log, policy_path, checkpoint_path


def run_multiprocess(cfg, checkpoint_dir, num_workers=3, verbose=True):
    """train_multiprocess.py:main (:423-470) on the drop-ins: `num_workers` spawned CPU processes step the environments, the
    learner serves them all from one batched forward per robot group (Collector.step_all / DQNPolicy.step_many), stores their
    transitions in the HBM rings, trains every `train_freq` collected steps, syncs the target nets and writes the checkpoints."""
    from simq.synth import synthetic_env_from_cfg
    num_robot_groups = len(cfg.robot_config)
    policy = simq.DQNPolicy(cfg, train=True, random_seed=cfg.seed)
    sgd = lambda net: torch.optim.SGD(net.parameters(), lr=cfg.learning_rate, momentum=0.9, weight_decay=cfg.weight_decay)
    optimizers = [sgd(n) for n in policy.policy_nets]
    replay_buffers = [simq.DeviceReplayBuffer(cfg.replay_buffer_size, cfg.num_input_channels) for _ in range(num_robot_groups)]
    start_timestep = 0
    if cfg.checkpoint_path is not None:
        start_timestep, _, replay_buffers = simq.resume(cfg.checkpoint_path, optimizers, cfg.num_input_channels, aliased=False)
    target_nets = policy.build_policy_nets()
    for i in range(num_robot_groups):
        target_nets[i].load_state_dict(policy.policy_nets[i].state_dict())
        target_nets[i].eval()
    learning_starts = int(round(cfg.learning_starts_frac * cfg.total_timesteps))
    total = learning_starts + cfg.total_timesteps
    collector = simq.Collector(cfg, policy, None, num_workers=num_workers, env_fn=synthetic_env_from_cfg)
    log, timestep, episodes = [], start_timestep, 0
    try:
        while timestep < total:
            eps = 1 - (1 - cfg.final_exploration) * min(1, max(0, timestep - learning_starts) / (cfg.exploration_frac * cfg.total_timesteps))
            for transitions_per_buffer, done in collector.step_all(eps):             # one env step of every worker
                for i, transitions in enumerate(transitions_per_buffer):              # Trainer.store_transitions
                    for transition in transitions:
                        replay_buffers[i].push(*transition)
                episodes += int(done)
                timestep += 1
                if timestep >= learning_starts and timestep % cfg.train_freq == 0:   # Trainer.step
                    for i in range(num_robot_groups):
                        if len(replay_buffers[i]) < cfg.batch_size:
                            continue
                        info = simq.train(cfg, policy.policy_nets[i], target_nets[i], optimizers[i], replay_buffers[i].sample(cfg.batch_size),
                                          policy.apply_transform, cfg.discount_factors[i])
                        log.append((timestep, i, info))
                        if verbose:
                            print('t=%d group %d %s' % (timestep, i, {k: round(v, 4) for k, v in info.items()}))
                if timestep % cfg.target_update_freq == 0:                            # Trainer.update_target_networks
                    for i in range(num_robot_groups):
                        target_nets[i].load_state_dict(policy.policy_nets[i].state_dict())
    finally:
        collector.close()
    policy_path = simq.save_policy(checkpoint_dir, timestep, policy.policy_nets)     # Trainer.save_checkpoint
    checkpoint_path = simq.save_checkpoint(checkpoint_dir, timestep, episodes, optimizers, replay_buffers)
    return policy, log, policy_path, checkpoint_path


def default_cfg(**over):
    cfg = types.SimpleNamespace(
        robot_config=[{'lifting_robot': 2}], num_input_channels=4, use_predicted_intention=False, use_predicted_intention_frac=0.5,
        final_exploration=0.01, exploration_frac=0.5, learning_starts_frac=0.25, total_timesteps=40, train_freq=2,
        target_update_freq=10, batch_size=8, replay_buffer_size=64, learning_rate=0.01, weight_decay=1e-4,
        grad_norm_clipping=100, use_double_dqn=True, discount_factors=[0.75, 0.75], checkpoint_path=None, policy_path=None, seed=0)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', type=int, default=0, help='collect with this many spawned environment processes (train_multiprocess.py)')
    ap.add_argument('--timesteps', type=int, default=40)
    ap.add_argument('--intention', action='store_true')
    ap.add_argument('--out', default='/tmp/simq_synthetic')
    a = ap.parse_args()
    c = default_cfg(total_timesteps=a.timesteps, use_predicted_intention=a.intention,
                    num_input_channels=5 if a.intention else 4,
                    robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}] if a.intention else [{'lifting_robot': 2}])
    _, lg, pp, cp = run_multiprocess(c, a.out, a.workers) if a.workers > 0 else run(c, a.out)
    print('%d training calls; saved %s and %s' % (len(lg), pp, cp))
