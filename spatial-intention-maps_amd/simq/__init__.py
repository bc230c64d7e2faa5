"""simq -- MI355X-native spatial-action-map DQN learner (hot path only).

Host-side mirror of the reference's Python interface for the path
train.py / networks.py / policies.py; the arithmetic lives in hand-written HIP
kernels behind the C-ABI of ``include/simq.h`` (libsimq.so).  Submodules that
need torch / the HIP library are imported lazily so that ``simq.arch`` and
``simq.synth`` stay importable on a bare CPU box.
"""
import importlib

_LAZY = {
    'FCN': ('.fcn', 'FCN'),
    'DQNPolicy': ('.policy', 'DQNPolicy'),
    'DQNIntentionPolicy': ('.policy', 'DQNIntentionPolicy'),
    'ReplayBuffer': ('.learner', 'ReplayBuffer'),
    'DeviceReplayBuffer': ('.learner', 'DeviceReplayBuffer'),
    'AliasedDeviceReplayBuffer': ('.learner', 'AliasedDeviceReplayBuffer'),
    'TransitionTracker': ('.tracker', 'TransitionTracker'),
    'Transition': ('.learner', 'Transition'),
    'train': ('.learner', 'train'),
    'train_step': ('.learner', 'train_step'),
    'train_groups': ('.learner', 'train_groups'),
    'StepOptions': ('.learner', 'StepOptions'),
    'learner_streams': ('.learner', 'learner_streams'),
    'train_step_dataparallel': ('.learner', 'train_step_dataparallel'),
    'train_intention': ('.learner', 'train_intention'),
    'train_intention_step': ('.learner', 'train_intention_step'),
    'lib': ('._lib', 'lib'),
    'Collector': ('.collector', 'Collector'),
    'CollectWorker': ('.collector', 'CollectWorker'),
    'save_policy': ('.checkpoint', 'save_policy'),
    'save_checkpoint': ('.checkpoint', 'save_checkpoint'),
    'load_checkpoint': ('.checkpoint', 'load_checkpoint'),
    'resume': ('.checkpoint', 'resume'),
}


def __getattr__(name):
    if name in _LAZY:
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod, __name__), attr)
    raise AttributeError(name)
