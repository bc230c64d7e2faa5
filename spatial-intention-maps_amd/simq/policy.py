"""simq.DQNPolicy -- drop-in for the reference ``policies.DQNPolicy`` (policies.py:11-74).

Same constructor, attributes and methods; each policy net is a simq.FCN (HIP) instead of
DataParallel(networks.FCN).  The three VectorEnv statics the reference calls are replaced by the
constants they return (simq.arch, citing envs.py:366-376,810,1090,2010) because the simulator is
out of scope.  `build_network` / `apply_transition` -- the names BASELINE.json uses -- are aliases
of the reference's real `build_policy_nets` / `apply_transform`.
"""
import random

import numpy as np
import torch

from . import arch
from ._lib import SimqError, lib, ptr, stream_ptr
from .fcn import FCN


class DQNPolicy:
    def __init__(self, cfg, train=False, random_seed=None):
        self.cfg = cfg
        self.robot_group_types = [next(iter(g.keys())) for g in self.cfg.robot_config]     # policies.py:14
        self.train = train
        if random_seed is not None:
            random.seed(random_seed)                                                        # policies.py:16-17
        self.num_robot_groups = len(self.robot_group_types)
        if not torch.cuda.is_available():
            raise SimqError('simq.DQNPolicy needs an MI355X (torch.cuda.is_available() is False); no CPU path')
        self.device = torch.device('cuda')                                                  # policies.py:21
        self.policy_nets = self.build_policy_nets()
        # Resume if applicable (policies.py:25-33)
        if getattr(self.cfg, 'checkpoint_path', None) is not None:
            self.policy_checkpoint = torch.load(self.cfg.policy_path, map_location=self.device)
            for i in range(self.num_robot_groups):
                self.policy_nets[i].load_state_dict(self.policy_checkpoint['state_dicts'][i])
                if self.train:
                    self.policy_nets[i].train()
                else:
                    self.policy_nets[i].eval()
            print("=> loaded policy '{}'".format(self.cfg.policy_path))

    def build_policy_nets(self):                                                            # policies.py:35-42
        policy_nets = []
        for robot_type in self.robot_group_types:
            num_output_channels = arch.get_num_output_channels(robot_type)
            policy_nets.append(FCN(num_input_channels=self.cfg.num_input_channels,
                                   num_output_channels=num_output_channels, device=self.device,
                                   precision=getattr(self.cfg, 'simq_precision', 'fp32')))
        return policy_nets

    build_network = build_policy_nets      # BASELINE.json's name for the same call

    def apply_transform(self, s):                                                           # policies.py:44-45
        """ToTensor on a float32 HWC ndarray (no scaling) + unsqueeze(0): [1,C,96,96]."""
        return torch.from_numpy(np.ascontiguousarray(s).transpose(2, 0, 1)).unsqueeze(0)

    apply_transition = apply_transform     # BASELINE.json's name for the same call

    def step(self, state, exploration_eps=None, debug=False):                               # policies.py:47-74
        if exploration_eps is None:
            exploration_eps = self.cfg.final_exploration
        action = [[None for _ in g] for g in state]
        output = [[None for _ in g] for g in state]
        with torch.no_grad():
            for i, g in enumerate(state):
                robot_type = self.robot_group_types[i]
                net = self.policy_nets[i]
                net.eval()
                # the reference runs one batch-1 forward per robot (HWC -> CHW -> device -> net -> argmax -> .cpu()); here
                # all robots of the group share ONE eval forward (eval-mode BatchNorm: samples are independent) and one
                # argmax launch; only the indices come back, the Q-maps only when `debug` asks for them.  The RNG draws
                # stay in the reference's (i, j) order.
                live = [j for j, s in enumerate(g) if s is not None]
                greedy, qmaps = net.infer_argmax_batch([g[j] for j in live], need_q=debug) if live else ([], [])
                for k, j in enumerate(live):
                    if random.random() < exploration_eps:
                        a = random.randrange(arch.get_action_space(robot_type))
                    else:
                        a = greedy[k]
                    action[i][j] = a
                    output[i][j] = qmaps[k] if debug else None
                if self.train:
                    net.train()
        if debug:
            info = {'output': output}
            return action, info
        return action


    def step_many(self, states, exploration_eps=None):
        """`step` for several environments at once (train_multiprocess.py:254-264 serves its collector workers
        one B=1 `policy.step` at a time): the awaiting robots of ALL environments share one eval forward + one argmax launch per
        robot group.  The epsilon-greedy draws are made in the order sequential `step(states[0])`, `step(states[1])`, ... calls
        would make them, so a seeded run selects the same actions.  Returns one action structure per environment."""
        if exploration_eps is None:
            exploration_eps = self.cfg.final_exploration
        actions = [[[None for _ in g] for g in st] for st in states]
        greedy = {}
        with torch.no_grad():
            for i in range(self.num_robot_groups):
                live = [(e, j) for e, st in enumerate(states) for j, s in enumerate(st[i]) if s is not None]
                if not live:
                    continue
                net = self.policy_nets[i]
                net.eval()
                idx, _ = net.infer_argmax_batch([states[e][i][j] for e, j in live], need_q=False)
                for (e, j), a in zip(live, idx):
                    greedy[(e, i, j)] = a
                if self.train:
                    net.train()
        for e, st in enumerate(states):                          # RNG draws: environment by environment, then (i, j) as in step()
            for i, g in enumerate(st):
                for j, s in enumerate(g):
                    if s is None:
                        continue
                    if random.random() < exploration_eps:
                        actions[e][i][j] = random.randrange(arch.get_action_space(self.robot_group_types[i]))
                    else:
                        actions[e][i][j] = greedy[(e, i, j)]
        return actions


class DQNIntentionPolicy(DQNPolicy):
    """Drop-in for policies.DQNIntentionPolicy (policies.py:76-146): one intention net FCN(C-1, 1) per robot group
    predicts the other robots' intention map, which is appended to the state before the Q-network runs.  The predicted
    map never leaves HBM between the two nets unless `debug` asks for it."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.intention_nets = self.build_intention_nets()
        if getattr(self.cfg, 'checkpoint_path', None) is not None:                          # policies.py:80-87
            for i in range(self.num_robot_groups):
                self.intention_nets[i].load_state_dict(self.policy_checkpoint['state_dicts_intention'][i])
                if self.train:
                    self.intention_nets[i].train()
                else:
                    self.intention_nets[i].eval()
            print("=> loaded intention network '{}'".format(self.cfg.policy_path))

    def build_intention_nets(self):                                                         # policies.py:89-95
        return [FCN(num_input_channels=self.cfg.num_input_channels - 1, num_output_channels=1, device=self.device,
                    precision=getattr(self.cfg, 'simq_precision', 'fp32')) for _ in range(self.num_robot_groups)]

    def _predict(self, i, s):
        """One HWC state [96,96,C-1] -> device tensors (state_with_intention [1,96,96,C], sigmoid map [96,96])."""
        net = self.intention_nets[i]
        x = torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)).unsqueeze(0).to(self.device)
        logit = net.forward_nhwc(x)
        C = x.shape[3]
        out = torch.empty((1, arch.STATE_WIDTH, arch.STATE_WIDTH, C + 1), dtype=torch.float32, device=self.device)
        prob = torch.empty((arch.STATE_WIDTH, arch.STATE_WIDTH), dtype=torch.float32, device=self.device)
        lib.call('simq_sigmoid_concat', ptr(x), ptr(logit), ptr(out), ptr(prob), arch.STATE_WIDTH * arch.STATE_WIDTH, C,
                 stream_ptr(self.device))
        return out, prob

    def step_intention(self, state, debug=False, _device=False):                            # policies.py:97-117
        state_intention = [[None for _ in g] for g in state]
        output_intention = [[None for _ in g] for g in state]
        with torch.no_grad():
            for i, g in enumerate(state):
                self.intention_nets[i].eval()
                for j, s in enumerate(g):
                    if s is not None:
                        out, prob = self._predict(i, s)
                        state_intention[i][j] = out if _device else out[0].cpu().numpy()
                        if debug:
                            output_intention[i][j] = prob.cpu().numpy()
                if self.train:
                    self.intention_nets[i].train()
        if debug:
            return state_intention, {'output_intention': output_intention}
        return state_intention

    def step(self, state, exploration_eps=None, debug=False, use_ground_truth_intention=False):   # policies.py:119-146
        if self.train and use_ground_truth_intention:
            return super().step(state, exploration_eps=exploration_eps, debug=debug)
        if self.train:                                                                      # drop the ground-truth map
            state = [[None if s is None else s[:, :, :-1] for s in g] for g in state]
        state = self.step_intention(state, debug=debug, _device=not debug)
        if debug:
            state, info_intention = state
        action = super().step(state, exploration_eps=exploration_eps, debug=debug)
        if debug:
            action, info = action
            info['state_intention'] = state
            info['output_intention'] = info_intention['output_intention']
            return action, info
        return action

    def step_many(self, states, exploration_eps=None, use_ground_truth_intention=False):
        """Several environments at once, predicted-intention path included: per robot group ONE batched forward of
        the intention net over the awaiting robots of all environments, one sigmoid+concat launch over the whole batch (the predicted
        maps stay in HBM), then the batched Q-network forward + argmax of DQNPolicy.step_many.  Same results and the same
        epsilon-greedy draw order as sequential step() calls (policies.py:119-146)."""
        if self.train and use_ground_truth_intention:
            return super().step_many(states, exploration_eps=exploration_eps)
        W = arch.STATE_WIDTH
        pred = [[[None for _ in g] for g in st] for st in states]
        with torch.no_grad():
            for i in range(self.num_robot_groups):
                live = [(e, j) for e, st in enumerate(states) for j, s in enumerate(st[i]) if s is not None]
                if not live:
                    continue
                net = self.intention_nets[i]
                net.eval()                                                                 # policies.py:101
                xs = [states[e][i][j] for e, j in live]
                if self.train:                                                             # drop the ground-truth map (policies.py:124)
                    xs = [s[:, :, :-1] for s in xs]
                x = torch.from_numpy(np.stack([np.ascontiguousarray(s, dtype=np.float32) for s in xs])).to(self.device)
                logit = net.forward_nhwc(x)
                n, C = x.shape[0], x.shape[3]
                out = torch.empty((n, W, W, C + 1), dtype=torch.float32, device=self.device)
                lib.call('simq_sigmoid_concat', ptr(x), ptr(logit), ptr(out), None, n * W * W, C, stream_ptr(self.device))
                for k, (e, j) in enumerate(live):
                    pred[e][i][j] = out[k:k + 1]
                if self.train:
                    net.train()
        return super().step_many(pred, exploration_eps=exploration_eps)
