"""Oracle: CPU restatement of the reference DQNPolicy.  TEST INFRASTRUCTURE.

Follows policies.py:11-146 literally, with the three VectorEnv statics it calls
replaced by the constants they return (below).  PINNED: oracle/gen_golden.py
imports the reference's own policies.py (torchvision.transforms.ToTensor and
envs.VectorEnv's three statics stood in for, since torchvision / pybullet are
absent here), runs policies.DQNPolicy / DQNIntentionPolicy on seeded weights and
asserts actions, RNG draw order and debug outputs bit-exact against the classes
below before it writes tests/golden/{policy,intention}_step.npz.
  envs.py:2010        Mapper.LOCAL_MAP_PIXEL_WIDTH = 96
  envs.py:810,1090    NUM_OUTPUT_CHANNELS: pushing 1; lifting/throwing/rescue 2
  envs.py:374-376     action space = channels * 96 * 96
"""
import random

import torch

from . import fcn
from .learner import apply_transform

STATE_WIDTH = 96
NUM_OUTPUT_CHANNELS = {'pushing_robot': 1, 'lifting_robot': 2, 'throwing_robot': 2, 'rescue_robot': 2}


def get_num_output_channels(robot_type):
    if robot_type not in NUM_OUTPUT_CHANNELS:
        raise Exception(robot_type)          # envs.py:1052
    return NUM_OUTPUT_CHANNELS[robot_type]


def get_action_space(robot_type):
    return get_num_output_channels(robot_type) * STATE_WIDTH * STATE_WIDTH


class DQNPolicy:
    """policies.py:11-74 over functional oracle states.  ``make_state(cin, cout)``
    supplies each group's initial state dict (the reference draws a fresh random
    init per build_policy_nets() call; fixtures pass seeded states instead)."""

    def __init__(self, cfg, make_state, train=False, random_seed=None):
        self.cfg = cfg
        self.robot_group_types = [next(iter(g.keys())) for g in self.cfg.robot_config]   # policies.py:14
        self.train = train
        if random_seed is not None:
            random.seed(random_seed)                                                      # policies.py:16-17
        self.num_robot_groups = len(self.robot_group_types)
        self._make_state = make_state
        self.policy_nets = self.build_policy_nets()

    def build_policy_nets(self):                                                          # policies.py:35-42
        nets = []
        for robot_type in self.robot_group_types:
            nets.append(self._make_state(self.cfg.num_input_channels, get_num_output_channels(robot_type)))
        return nets

    def apply_transform(self, s):                                                         # policies.py:44-45
        return apply_transform(s)

    def step(self, state, exploration_eps=None, debug=False):                             # policies.py:47-74
        if exploration_eps is None:
            exploration_eps = self.cfg.final_exploration
        action = [[None for _ in g] for g in state]
        output = [[None for _ in g] for g in state]
        with torch.no_grad():
            for i, g in enumerate(state):
                robot_type = self.robot_group_types[i]
                for j, s in enumerate(g):
                    if s is not None:
                        s = self.apply_transform(s)
                        o = fcn.fcn_forward(self.policy_nets[i], s, False).squeeze(0)     # eval mode, :56,:60
                        if random.random() < exploration_eps:                             # :61
                            a = random.randrange(get_action_space(robot_type))            # :62
                        else:
                            a = o.view(1, -1).max(1)[1].item()                            # :64
                        action[i][j] = a
                        output[i][j] = o.cpu().numpy()
        if debug:
            return action, {'output': output}
        return action


class DQNIntentionPolicy(DQNPolicy):
    """policies.py:76-146 over functional oracle states (intention nets: FCN(num_input_channels - 1, 1))."""

    def __init__(self, cfg, make_state, train=False, random_seed=None):
        super().__init__(cfg, make_state, train=train, random_seed=random_seed)
        self.intention_nets = self.build_intention_nets()                                 # policies.py:79

    def build_intention_nets(self):                                                       # policies.py:89-95
        return [self._make_state(self.cfg.num_input_channels - 1, 1) for _ in range(self.num_robot_groups)]

    def step_intention(self, state, debug=False):                                         # policies.py:97-117
        import numpy as np
        state_intention = [[None for _ in g] for g in state]
        output_intention = [[None for _ in g] for g in state]
        with torch.no_grad():
            for i, g in enumerate(state):
                for j, s in enumerate(g):
                    if s is not None:
                        s_copy = s.copy()
                        s = self.apply_transform(s)
                        o = torch.sigmoid(fcn.fcn_forward(self.intention_nets[i], s, False)).squeeze(0).squeeze(0).numpy()
                        state_intention[i][j] = np.concatenate((s_copy, np.expand_dims(o, 2)), axis=2)
                        output_intention[i][j] = o
        if debug:
            return state_intention, {'output_intention': output_intention}
        return state_intention

    def step(self, state, exploration_eps=None, debug=False, use_ground_truth_intention=False):   # policies.py:119-146
        if self.train and use_ground_truth_intention:
            return super().step(state, exploration_eps=exploration_eps, debug=debug)
        if self.train:
            state = [[None if s is None else s[:, :, :-1] for s in g] for g in state]     # drop the ground-truth map
        state = self.step_intention(state, debug=debug)
        if debug:
            state, info_intention = state
        action = super().step(state, exploration_eps=exploration_eps, debug=debug)
        if debug:
            action, info = action
            info['state_intention'] = state
            info['output_intention'] = info_intention['output_intention']
            return action, info
        return action
