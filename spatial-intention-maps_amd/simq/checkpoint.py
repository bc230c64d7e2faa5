"""Checkpoint files of the reference trainer, read and written from the device-resident learner.

Reference behaviour mirrored:
  write   train.py:309-346   policy_{t:08d}.pth.tar    {'timestep', 'state_dicts'[, 'state_dicts_intention']}
                             checkpoint_{t:08d}.pth.tar {'timestep', 'episode', 'optimizers', 'replay_buffers'
                                                         [, 'optimizers_intention']}; older checkpoint_* files are removed
  read    train.py:197-209   optimizers[i].load_state_dict(...), replay_buffers[i] = checkpoint['replay_buffers'][i]
          policies.py:25-33, 80-87   (policy files: handled by DQNPolicy / DQNIntentionPolicy themselves)

The replay rings are stored in the reference's own host layout (a ReplayBuffer whose .buffer is a list of Transition
records holding float32 HWC ndarrays, None for terminal next states); an observation shared by two transitions is ONE
ndarray object, so the pickle holds it once -- the same aliasing the collector produces (train.py:61-66).

Class names in the pickle, both directions: the reference's scripts run as `__main__` (`python train.py`), so THEIR files name the
ring classes `__main__.ReplayBuffer` / `__main__.Transition` (`train.*` / `train_multiprocess.*` when imported as modules);
load_checkpoint resolves all of those to this package's classes, so such a file loads without the reference on the path.
save_checkpoint writes the rings under `__main__.ReplayBuffer` / `__main__.Transition` too (reference_names=True, the default):
the reference's own `torch.load` (train.py:200) then rebuilds them as ITS classes -- same attributes (capacity, buffer, position;
state, action, reward, next_state) -- without this package being importable.  reference_names=False keeps `simq.learner.*`.
"""
import glob
import os
import pickle

import numpy as np
import torch

from ._lib import SimqError
from . import learner as _learner

_REFERENCE_CLASSES = {('train', 'ReplayBuffer'): 'ReplayBuffer', ('train', 'Transition'): 'Transition',
                      ('train_multiprocess', 'ReplayBuffer'): 'ReplayBuffer', ('train_multiprocess', 'Transition'): 'Transition',
                      ('__main__', 'ReplayBuffer'): 'ReplayBuffer', ('__main__', 'Transition'): 'Transition'}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        mapped = _REFERENCE_CLASSES.get((module, name))
        if mapped is not None:
            return getattr(_learner, mapped)
        return super().find_class(module, name)


class _ReferenceNamesPickler(pickle._Pickler):
    """Writes this package's ReplayBuffer / Transition CLASSES as the globals `__main__.ReplayBuffer` / `__main__.Transition`
    (the names in a file written by `python train.py`); everything else as usual.  (The pure-Python pickler: the C one has no
    hook for how a class reference is written.  Observations are bulk `bytes`, which it writes through unchanged.)"""

    def save_global(self, obj, name=None):
        alias = {_learner.ReplayBuffer: b'ReplayBuffer', _learner.Transition: b'Transition'}.get(obj) if isinstance(obj, type) else None
        if alias is None:
            return super().save_global(obj, name)
        self.write(pickle.GLOBAL + b'__main__\n' + alias + b'\n')
        self.memoize(obj)


class _PickleModule:
    """What torch.load / torch.save want from `pickle_module`: the pickle namespace with our Unpickler (and plain Pickler)."""
    __name__ = 'pickle'
    Unpickler = _Unpickler
    Pickler = pickle.Pickler
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()

    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    loads = staticmethod(pickle.loads)


def _f32(obs):
    """The observation as a float32 ndarray WITHOUT changing its identity when it already is one (np.asarray with a dtype
    hands back a fresh view object, which would defeat the aliased ring's one-upload-per-object rule)."""
    if obs is None or (isinstance(obs, np.ndarray) and obs.dtype == np.float32):
        return obs
    return np.asarray(obs, dtype=np.float32)


def to_host_ring(buf):
    """Device ring (DeviceReplayBuffer / AliasedDeviceReplayBuffer) -> host ReplayBuffer in the reference's layout.
    Slots referenced by several transitions come back as one ndarray object (one D2H copy, one pickle entry)."""
    if isinstance(buf, _learner.ReplayBuffer):
        return buf
    host = _learner.ReplayBuffer(buf.capacity)
    host.position = buf.position
    cache = {}
    # the ring's slots are written by asynchronous H2D copies on the upload stream (or on whichever stream pushed): the D2H copies below run
    # on the current stream and are ordered behind none of them -- wait until every push so far has landed
    buf.sync_ring()

    def fetch(obs):
        if obs is None:
            return None
        key = (obs.store.data_ptr(), int(obs))
        if key not in cache:
            cache[key] = obs.store[int(obs)].cpu().numpy()
        return cache[key]

    for rec in buf.buffer:
        host.buffer.append(_learner.Transition(fetch(rec.state), rec.action, rec.reward, fetch(rec.next_state)))
    return host


def to_device_ring(host, num_input_channels=None, device=None, aliased=True, pool_slots=None):
    """Host ReplayBuffer (ours or one unpickled from a reference checkpoint) -> device ring with the same
    capacity / position / record order, hence the same `random.sample` picks."""
    if isinstance(host, _learner.DeviceReplayBuffer):
        return host
    if num_input_channels is None:
        if not host.buffer:
            raise SimqError('to_device_ring: an empty ring does not tell the channel count; pass num_input_channels')
        num_input_channels = np.asarray(host.buffer[0].state).shape[-1]
    if aliased:
        if pool_slots is None:
            distinct = {id(o) for r in host.buffer for o in (r.state, r.next_state) if o is not None}
            pool_slots = max(host.capacity + max(256, host.capacity // 4), len(distinct) + 256)
        dev = _learner.AliasedDeviceReplayBuffer(host.capacity, num_input_channels, device=device, pool_slots=pool_slots)
    else:
        dev = _learner.DeviceReplayBuffer(host.capacity, num_input_channels, device=device)
    n = len(host.buffer)
    full = n == host.capacity
    order = list(range(host.position, n)) + list(range(host.position)) if full else list(range(n))
    if not full and host.position != n:
        raise SimqError('to_device_ring: ring with %d records but position %d' % (n, host.position))
    for i in order:                                   # oldest first, so the ring ends at host.position again
        rec = host.buffer[i]
        if full and dev.position != i:
            dev.position = i
            while len(dev.buffer) < i:
                dev.buffer.append(None)
        dev.push(_f32(rec.state), rec.action, rec.reward, _f32(rec.next_state))
    if dev.position != host.position or len(dev.buffer) != n:
        raise SimqError('to_device_ring: rebuilt ring is at position %d / %d records, expected %d / %d'
                        % (dev.position, len(dev.buffer), host.position, n))
    return dev


def save_policy(checkpoint_dir, timestep, policy_nets, intention_nets=None):
    """train.py:314-322."""
    os.makedirs(str(checkpoint_dir), exist_ok=True)
    path = os.path.join(str(checkpoint_dir), 'policy_{:08d}.pth.tar'.format(timestep))
    out = {'timestep': timestep, 'state_dicts': [n.state_dict() for n in policy_nets]}
    if intention_nets is not None:
        out['state_dicts_intention'] = [n.state_dict() for n in intention_nets]
    torch.save(out, path)
    return path


class _ReferenceNamesPickleModule(_PickleModule):
    Pickler = _ReferenceNamesPickler


def save_checkpoint(checkpoint_dir, timestep, episode, optimizers, replay_buffers, optimizers_intention=None,
                    remove_old=True, reference_names=True):
    """train.py:324-345: optimizer state + replay rings; older checkpoint_* files in the directory are removed.
    reference_names: the ring classes are written as `__main__.ReplayBuffer` / `__main__.Transition`, i.e. exactly what the
    reference's scripts write and can read back (module docstring)."""
    os.makedirs(str(checkpoint_dir), exist_ok=True)
    path = os.path.join(str(checkpoint_dir), 'checkpoint_{:08d}.pth.tar'.format(timestep))
    out = {'timestep': timestep, 'episode': episode,
           'optimizers': [o.state_dict() for o in optimizers],
           'replay_buffers': [to_host_ring(b) for b in replay_buffers]}
    if optimizers_intention is not None:
        out['optimizers_intention'] = [o.state_dict() for o in optimizers_intention]
    if reference_names:
        torch.save(out, path, pickle_module=_ReferenceNamesPickleModule)
    else:
        torch.save(out, path)
    if remove_old:
        for old in glob.glob(os.path.join(str(checkpoint_dir), 'checkpoint_*.pth.tar')):
            if os.path.abspath(old) != os.path.abspath(path):
                os.unlink(old)
    return path


def load_checkpoint(path, map_location=None):
    """torch.load of a checkpoint_*.pth.tar written by save_checkpoint or by the reference trainer (train.py:324-334).
    The file pickles Python objects (the replay rings), so it is read with weights_only=False: load only files you wrote
    or trust, as with the reference's own torch.load (train.py:200)."""
    return torch.load(str(path), map_location=map_location, pickle_module=_PickleModule, weights_only=False)


def resume(checkpoint, optimizers, num_input_channels=None, device=None, optimizers_intention=None, aliased=True):
    """train.py:200-208 on the device path: restores the optimizers in place and returns
    (start_timestep, episode, replay_buffers) with the rings uploaded into HBM."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = load_checkpoint(checkpoint)
    if len(checkpoint['optimizers']) != len(optimizers):
        raise SimqError('resume: checkpoint holds %d optimizers, the trainer has %d robot groups'
                        % (len(checkpoint['optimizers']), len(optimizers)))
    for opt, sd in zip(optimizers, checkpoint['optimizers']):
        opt.load_state_dict(sd)
    if optimizers_intention is not None:
        for opt, sd in zip(optimizers_intention, checkpoint['optimizers_intention']):
            opt.load_state_dict(sd)
    rings = [to_device_ring(b, num_input_channels, device=device, aliased=aliased) for b in checkpoint['replay_buffers']]
    return checkpoint['timestep'], checkpoint['episode'], rings
