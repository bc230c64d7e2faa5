"""TransitionTracker (train.py:47-68), free of torch / HIP imports so that collector worker processes stay light."""


class TransitionTracker:
    """Drop-in for train.TransitionTracker (train.py:47-68): remembers, per robot, the observation and action the robot is
    waiting on and turns (reward, new observation, done) into replay transitions.  The SAME ndarray object is handed out
    as `next_state` of one transition and `state` of the next one -- AliasedDeviceReplayBuffer uploads it once."""

    def __init__(self, initial_state):
        self.num_buffers = len(initial_state)
        self.prev_state = initial_state
        self.prev_action = [[None] * len(group) for group in initial_state]

    def update_action(self, action):
        for i, group in enumerate(action):
            for j, a in enumerate(group):
                if a is not None:
                    self.prev_action[i][j] = a

    def update_step_completed(self, reward, state, done):
        out = [[] for _ in range(self.num_buffers)]
        for i, group in enumerate(state):
            for j, s in enumerate(group):
                if s is None and not done:
                    continue                                    # this robot has not finished its action yet
                before = self.prev_state[i][j]
                if before is not None:
                    out[i].append((before, self.prev_action[i][j], reward[i][j], s))
                self.prev_state[i][j] = s
        return out
