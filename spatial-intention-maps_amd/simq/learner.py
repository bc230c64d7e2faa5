"""DQN learner: drop-in for the learner half of the reference train.py.

Reference symbols mirrored (same names, argument meaning, return values):
  Transition               train.py:26
  ReplayBuffer             train.py:28-45   (host python-list ring, global `random` stream)
  train(cfg, policy_net, target_net, optimizer, batch, transform_fn, discount_factor)
                           train.py:108-141 -> {'td_error': float, 'loss': float}
  train_intention(intention_net, optimizer, batch, transform_fn)
                           train.py:143-158 -> {'loss_intention': float}
  TransitionTracker        train.py:47-68
Additions for the MI355X path:
  DeviceReplayBuffer       same push/sample/len contract, states live in an HBM ring
                           ([capacity][96][96][C] fp32, the reference's own HWC layout) and
                           sample() is an index gather driven by the same random.sample picks
  AliasedDeviceReplayBuffer   the same with every observation stored once (next_state of t == state of t+1)
  train_step(...)          the fused step over flat buffers (no torch autograd), used by train()
All arithmetic goes through libsimq (HIP); there is no torch/CPU fallback.
"""
import ctypes
import os
import random
from collections import namedtuple

import numpy as np
import torch

from . import arch, dist as sdist
from ._lib import MODE_EVAL, MODE_TRAIN, MODE_TRAIN_NOGRAD, SimqError, TrainArgs, lib, ptr, stream_ptr
from .fcn import FCN
from .tracker import TransitionTracker  # noqa: F401  (train.py:47-68; lives in its own torch-free module for the collector workers)

Transition = namedtuple('Transition', ('state', 'action', 'reward', 'next_state'))   # train.py:26
W = arch.STATE_WIDTH


class ReplayBuffer:
    """train.py:28-45 (host ring; kept for drop-in use and checkpoint compatibility)."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.buffer = []
        self.position = 0

    def push(self, *args):
        if len(self.buffer) < self.capacity:
            self.buffer.append(None)
        self.buffer[self.position] = Transition(*args)
        self.position = (self.position + 1) % self.capacity

    def sample(self, batch_size):
        transitions = random.sample(self.buffer, batch_size)
        return Transition(*zip(*transitions))

    def __len__(self):
        return len(self.buffer)


class DeviceBatch:
    """A sampled minibatch already resident in HBM (what DeviceReplayBuffer.sample returns).

    state [B,96,96,C] f32 NHWC, action [B] i64, reward [B] f32 (device);
    next_state [N',96,96,C] (non-final only), nonfinal_pos [N'] i32 device, non_final_mask host bool list.
    """
    __slots__ = ('state', 'action', 'reward', 'next_state', 'nonfinal_pos', 'non_final_mask', 'ready_event')

    def __init__(self, state, action, reward, next_state, nonfinal_pos, non_final_mask, ready_event=None):
        self.state, self.action, self.reward = state, action, reward
        self.next_state, self.nonfinal_pos, self.non_final_mask = next_state, nonfinal_pos, non_final_mask
        # set when the tensors were produced on a stream of their own (DeviceReplayBuffer.gather): the event behind their last writer.  The
        # consuming stream has already been made to wait for it; a stream that wants to start EARLIER (the target-net forward of the next
        # step beside the running step, StepOptions.early_target_forward) waits for it itself.
        self.ready_event = ready_event


def assemble_batch(batch, device, allow_all_final=False):
    """train.py:109-112,116-117: host Transition-of-tuples -> DeviceBatch.  The reference
    transposes every HWC state to CHW and concatenates; the HIP path consumes HWC directly, so
    this is one stack + one H2D copy per tensor.  allow_all_final: a data-parallel SHARD may consist of terminal
    transitions only (the reference fails only when the WHOLE minibatch does); its next_state tensor is then empty."""
    if isinstance(batch, DeviceBatch):
        return batch
    state = torch.from_numpy(np.stack(batch.state)).to(device, non_blocking=True)
    action = torch.tensor(batch.action, dtype=torch.long).to(device, non_blocking=True)
    reward = torch.tensor(batch.reward, dtype=torch.float32).to(device, non_blocking=True)
    mask = [s is not None for s in batch.next_state]
    nf = [s for s in batch.next_state if s is not None]
    if not nf and not allow_all_final:
        raise SimqError('train: batch has no non-final next state (the reference raises at train.py:112 too)')
    if nf:
        next_state = torch.from_numpy(np.stack(nf)).to(device, non_blocking=True)
    else:
        next_state = torch.empty((0,) + tuple(state.shape[1:]), dtype=torch.float32, device=device)
    pos = torch.tensor([i for i, m in enumerate(mask) if m], dtype=torch.int32).to(device, non_blocking=True)
    return DeviceBatch(state, action, reward, next_state, pos, mask)


_PACK_RINGS = {}        # device -> [pinned uint8 buffers, events, position]
_UPLOAD_STREAMS = {}    # device -> the stream the packed per-batch uploads (and, by default, the ring's pushes and gathers) run on


def _upload_stream(device):
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    up = _UPLOAD_STREAMS.get(device)
    if up is None:
        # (a stream TESTED not to share the hardware queue of the stream the rings are used from -- behind a learner's launch stream the
        # next minibatch's gathers would wait for the whole running step: _HardwareQueues further down)
        q = _hardware_queues(device)
        up = _UPLOAD_STREAMS[device] = q.acquire({q.classify(torch.cuda.current_stream(device))} if q.ok else set(), 1)[0]
    return up


def sync_uploads(device):
    """Host-side wait for everything issued on `device`'s upload stream (ring pushes, index uploads, gathers): what a reader of the ring
    on ANOTHER stream (a checkpoint's D2H copies, `np.asarray(record.state)`) needs before it may look at a slot."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    up = _UPLOAD_STREAMS.get(device)
    if up is not None:
        up.synchronize()


class StepOptions(namedtuple('StepOptions', ('fused', 'overlap_target_forward', 'early_target_forward'))):
    """How ONE learner issues its TD step -- a property of the call (train_step(options=...)) or of the learner (FCN.step_options, which
    the drop-in train() reads), never of the process (round 6: the module-level A/B switches of rounds 3-5 are gone, as the library's
    simq_tune_* setters went in round 5).
      fused                    the step is ONE library call (simq_train_step) instead of ~15 ctypes calls issued from Python
      overlap_target_forward   the (independent) target-net forward on the learner's side stream (False: everything on the launch stream --
                               bench.py's per-kernel roofline pass, where concurrent kernels would share the device)
      early_target_forward     the target-net forward of a step on a stream of its own that does not wait for the previous step (needs a
                               minibatch gathered on the upload stream: DeviceBatch.ready_event)"""
    __slots__ = ()

    def __new__(cls, fused=True, overlap_target_forward=True, early_target_forward=True):
        return super().__new__(cls, bool(fused), bool(overlap_target_forward), bool(early_target_forward))


DEFAULT_STEP_OPTIONS = StepOptions()


def _upload_packed(device, arrays, slots=4, upload_stream=True):
    """numpy arrays -> device tensors of the same dtypes through ONE asynchronous H2D copy out of a small ring of pinned host buffers
    (a buffer is reused only after the copy that last read it has finished).  Non-CUDA devices: plain copies.
    upload_stream=False: the copy on the consuming stream (A/B of rounds 1-3's form)."""
    if device.type != 'cuda':
        return tuple(torch.from_numpy(a).to(device) for a in arrays)
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    total = max(total, 16)
    ring = _PACK_RINGS.get(device)
    if ring is None or ring[0][0].numel() < total:
        ring = _PACK_RINGS[device] = [[torch.empty(max(total, 1 << 16), dtype=torch.uint8).pin_memory() for _ in range(slots)], [None] * slots, 0]
    bufs, events, i = ring
    ring[2] = (i + 1) % slots
    if events[i] is not None:
        events[i].synchronize()
    host = bufs[i].numpy()
    for a, o in zip(arrays, offs):
        if a.size:
            host[o:o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    main = torch.cuda.current_stream(device)
    if upload_stream:
        # the copy depends on nothing the device is doing: on its own stream it runs while the previous step's kernels still do, and the
        # consuming stream only waits for its event (on the consuming stream it queued behind the whole previous step and the device idled
        # for the copy's latency at every step boundary).  The buffer comes from the upload stream's pool and is handed to the consumer.
        up = _upload_stream(device)
        with torch.cuda.stream(up):
            devbuf = torch.empty(total, dtype=torch.uint8, device=device)
            devbuf.copy_(bufs[i][:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up)
        main.wait_event(ev)
        devbuf.record_stream(main)
    else:
        devbuf = torch.empty(total, dtype=torch.uint8, device=device)
        devbuf.copy_(bufs[i][:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
    events[i] = ev
    kinds = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32}
    return tuple(devbuf[o:o + a.nbytes].view(kinds[a.dtype]) for a, o in zip(arrays, offs))


class _DeviceObs:
    """One observation resident in an HBM ring slot, standing where the reference keeps the ndarray itself
    (`random.choice(replay_buffers[i].buffer).state`, train.py:294): converts to the [96,96,C] float32 array on demand
    (np.asarray / slicing), and to its slot number for the index gathers."""
    __slots__ = ('store', 'slot')

    def __init__(self, store, slot):
        self.store, self.slot = store, int(slot)

    def __index__(self):
        return self.slot

    __int__ = __index__

    @property
    def shape(self):
        return tuple(self.store.shape[1:])

    def __array__(self, dtype=None, copy=None):
        if self.store.is_cuda:
            sync_uploads(self.store.device)    # (ring slots are written on the upload stream: _PinnedStaging.upload)
        a = self.store[self.slot].cpu().numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, key):
        return self.__array__()[key]

    def copy(self):
        return self.__array__()

    def __repr__(self):
        return '_DeviceObs(slot=%d, shape=%s)' % (self.slot, self.shape)


class _PinnedStaging:
    """Host side of the collector hand-off: observations are copied into a small ring of pinned host slots and
    uploaded with asynchronous H2D copies, so `push` returns as soon as the ndarray is staged and the environment can keep stepping
    while the copy (and the learner's kernels) run.  A slot is reused only after the copy that last read it has finished."""

    def __init__(self, item_shape, device, slots=32, on_upload_stream=True):
        self.device = device
        self.on_upload_stream = bool(on_upload_stream)
        self.enabled = device.type == 'cuda' and torch.cuda.is_available()
        self.slots = slots
        self.pos = 0
        if self.enabled:
            self.buf = torch.empty((slots,) + tuple(item_shape), dtype=torch.float32).pin_memory()
            self.events = [None] * slots

    def upload(self, arr, dst):
        if not self.enabled:
            dst.copy_(torch.as_tensor(arr))
            return
        i = self.pos
        self.pos = (i + 1) % self.slots
        if self.events[i] is not None:
            self.events[i].synchronize()                       # the copy that last used this slot (32 pushes ago) is long done
        self.buf[i].copy_(torch.as_tensor(arr))                # host memcpy into pinned memory
        # asynchronous H2D -- on the UPLOAD stream when the gathers run there (the ring's `upload_stream` option): the ring is then written
        # and read on one stream (stream order is the only ordering needed) and neither a push nor the next gather queues behind the
        # learner's running step on the consuming stream.  Otherwise on the current stream, with an event for a gather elsewhere to wait on.
        up = _upload_stream(self.device) if self.on_upload_stream else None
        if up is not None:
            with torch.cuda.stream(up):
                dst.copy_(self.buf[i], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(up)
            self.last_event = None
        else:
            dst.copy_(self.buf[i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.last_event = ev                               # (a gather on another stream orders itself behind the ring's last write)
        self.events[i] = ev


class DeviceReplayBuffer:
    """ReplayBuffer (train.py:28-45) with the states in an HBM ring.

    push(state, action, reward, next_state) / sample(B) / len() / .position as in the reference;
    `.buffer` is a list of Transition records whose state fields are _DeviceObs handles (ring slot +
    on-demand ndarray view, so `random.choice(buf.buffer).state` of train.py:294 still yields the observation).  Sampling draws `random.sample(range(len), B)` -- the same picks the reference's
    `random.sample(self.buffer, B)` makes under the same seed (golden: tests/golden/sampler.npz).
    """

    def __init__(self, capacity, num_input_channels, device=None, upload_stream=True):
        self.capacity = int(capacity)
        self.C = int(num_input_channels)
        self.device = torch.device('cuda' if device is None else device)
        self.item = W * W * self.C
        self.states = torch.empty((self.capacity, W, W, self.C), dtype=torch.float32, device=self.device)
        self.next_states = torch.empty((self.capacity, W, W, self.C), dtype=torch.float32, device=self.device)
        self._set_upload_mode(upload_stream)
        self._staging = _PinnedStaging((W, W, self.C), self.device, on_upload_stream=self.ring_on_upload_stream)
        self.buffer = []
        self.position = 0

    def _set_upload_mode(self, upload_stream):
        """upload_stream: True (default) -- pushes, the per-batch index upload and the gathers all run on the device's upload stream, beside
        whatever the learner has queued on the consuming stream (a gathered DeviceBatch then carries `ready_event`); 'index' -- only the
        index upload does, the gathers run on the consuming stream (round 4's form); False -- everything on the consuming stream.  A
        property of THIS ring (round 6; it was a module-level switch)."""
        if upload_stream not in (True, False, 'index'):
            raise SimqError("DeviceReplayBuffer: upload_stream must be True, False or 'index' (got %r)" % (upload_stream,))
        self.upload_stream = upload_stream
        self.ring_on_upload_stream = upload_stream is True and self.device.type == 'cuda'

    def sync_ring(self):
        """Host-side wait until every push so far has landed in HBM (the H2D copies run on the upload stream / the pushing stream):
        for readers that look at ring slots outside stream order -- checkpoint.to_host_ring, np.asarray(record.state)."""
        if self.device.type != 'cuda':
            return
        sync_uploads(self.device)
        for ev in (getattr(self, '_bulk_event', None), getattr(getattr(self, '_staging', None), 'last_event', None)):
            if ev is not None:
                ev.synchronize()

    def push(self, state, action, reward, next_state):
        if len(self.buffer) < self.capacity:
            self.buffer.append(None)
        slot = self.position
        self._staging.upload(state, self.states[slot])
        if next_state is not None:
            self._staging.upload(next_state, self.next_states[slot])
        self.buffer[slot] = Transition(_DeviceObs(self.states, slot), int(action), float(reward),
                                       _DeviceObs(self.next_states, slot) if next_state is not None else None)
        self.position = (self.position + 1) % self.capacity

    def push_many(self, states, actions, rewards, next_states, terminal):
        """Bulk fill (bench / tests): arrays [n,96,96,C], [n], [n], [n,96,96,C], bool [n]."""
        n = len(actions)
        if len(self.buffer) + n > self.capacity or self.position != len(self.buffer):
            for i in range(n):
                self.push(states[i], actions[i], rewards[i], None if terminal[i] else next_states[i])
            return
        s0 = self.position
        self.states[s0:s0 + n].copy_(torch.as_tensor(states))
        self.next_states[s0:s0 + n].copy_(torch.as_tensor(next_states))
        if self.device.type == 'cuda':
            self._bulk_event = torch.cuda.Event()
            self._bulk_event.record(torch.cuda.current_stream(self.device))
        for i in range(n):
            self.buffer.append(Transition(_DeviceObs(self.states, s0 + i), int(actions[i]), float(rewards[i]),
                                          None if terminal[i] else _DeviceObs(self.next_states, s0 + i)))
        self.position = (s0 + n) % self.capacity

    def __len__(self):
        return len(self.buffer)

    def sample_indices(self, batch_size):
        return random.sample(range(len(self.buffer)), batch_size)

    def gather(self, idx, allow_all_final=False):
        """allow_all_final: data-parallel ranks gather only their slice of the drawn indices, and a slice may hold terminal
        transitions only (the minibatch as a whole is checked by the caller / fails as the reference does)."""
        recs = [self.buffer[i] for i in idx]
        B = len(recs)
        dev = self.device
        st = stream_ptr(dev)
        mask = [r.next_state is not None for r in recs]
        nf = [int(r.next_state) for r in recs if r.next_state is not None]
        if not nf and not allow_all_final:
            raise SimqError('sample: no non-final next state in the batch (train.py:112 would raise)')
        # the five small per-batch arrays travel as ONE asynchronous copy out of pinned memory: a copy from pageable memory makes the
        # host wait for everything already on the stream -- the whole previous step -- and the device then idles while the host
        # issues the rest (kernel trace: a 70-80 us hole between two copy kernels at the start of every step)
        index, nindex, action, reward, pos = _upload_packed(dev, (
            np.asarray([int(r.state) for r in recs], np.int64), np.asarray(nf, np.int64),
            np.asarray([r.action for r in recs], np.int64), np.asarray([r.reward for r in recs], np.float32),
            np.asarray([i for i, m in enumerate(mask) if m], np.int32)), upload_stream=self.upload_stream is not False)
        up = _upload_stream(dev) if self.ring_on_upload_stream else None
        if up is not None:
            # the two gathers on the upload stream, behind the index copy they read: they depend on the ring and the indices only, not on the
            # step that is still running on the consuming stream -- the minibatch is ready while that step runs, and a consumer that can start
            # early (StepOptions.early_target_forward) finds it there.  The ring's last push (a copy on the consuming stream) is waited for first.
            main = torch.cuda.current_stream(dev)
            ev_push = getattr(self._staging, 'last_event', None) if hasattr(self, '_staging') else None
            if getattr(self, '_bulk_event', None) is not None:
                up.wait_event(self._bulk_event)
            if ev_push is not None:
                up.wait_event(ev_push)
            with torch.cuda.stream(up):
                state = torch.empty((B, W, W, self.C), dtype=torch.float32, device=dev)
                lib.call('simq_replay_gather', ptr(self.states), self.item, ptr(index), B, ptr(state), stream_ptr(dev))
                next_state = torch.empty((len(nf), W, W, self.C), dtype=torch.float32, device=dev)
                if nf:
                    lib.call('simq_replay_gather', ptr(self.next_states), self.item, ptr(nindex), len(nf), ptr(next_state), stream_ptr(dev))
                ready = torch.cuda.Event()
                ready.record(up)
            main.wait_event(ready)
            for t in (state, next_state, index, nindex):
                t.record_stream(main)
            return DeviceBatch(state, action, reward, next_state, pos, mask, ready)
        state = torch.empty((B, W, W, self.C), dtype=torch.float32, device=dev)
        lib.call('simq_replay_gather', ptr(self.states), self.item, ptr(index), B, ptr(state), st)
        next_state = torch.empty((len(nf), W, W, self.C), dtype=torch.float32, device=dev)
        if nf:
            lib.call('simq_replay_gather', ptr(self.next_states), self.item, ptr(nindex), len(nf), ptr(next_state), st)
        return DeviceBatch(state, action, reward, next_state, pos, mask)

    def sample(self, batch_size):
        return self.gather(self.sample_indices(batch_size))


class AliasedDeviceReplayBuffer(DeviceReplayBuffer):
    """DeviceReplayBuffer whose observations live ONCE in HBM: the collector hands the same ndarray over
    as `next_state` of transition t and `state` of transition t+1 (train.py:61-66, 241-244), so each 96x96xC observation is
    uploaded once into an observation pool and transitions hold two pool slots.  Half the HBM and half the H2D traffic
    of the two-ring layout; same push / sample / len contract and the same `random.sample` picks.

    pool_slots: observations the pool can hold (default capacity + 25 %, enough when pushes alias as the collector's do;
    independent state / next_state arrays need up to 2 * capacity)."""

    def __init__(self, capacity, num_input_channels, device=None, pool_slots=None, upload_stream=True):
        self.capacity = int(capacity)
        self.C = int(num_input_channels)
        self.device = torch.device('cuda' if device is None else device)
        self.item = W * W * self.C
        n = int(pool_slots) if pool_slots is not None else self.capacity + max(256, self.capacity // 4)
        self.pool = torch.empty((n, W, W, self.C), dtype=torch.float32, device=self.device)
        self.states = self.next_states = self.pool          # both gathers of the base class read the one pool
        self._set_upload_mode(upload_stream)
        self._staging = _PinnedStaging((W, W, self.C), self.device, on_upload_stream=self.ring_on_upload_stream)
        self._free = list(range(n - 1, -1, -1))
        self._ref = [0] * n
        self._recent = {}                                   # id(ndarray) -> (slot, ndarray): observations uploaded lately
        self._recent_order = []
        self.buffer = []
        self.position = 0

    def _upload(self, arr):
        hit = self._recent.get(id(arr))
        if hit is not None and hit[1] is arr:
            return hit[0]
        if not self._free:
            raise SimqError('AliasedDeviceReplayBuffer: observation pool exhausted (%d slots); the pushes do not alias '
                            'next_state/state -- construct with pool_slots=2*capacity' % len(self._ref))
        slot = self._free.pop()
        self._staging.upload(arr, self.pool[slot])
        self._recent[id(arr)] = (slot, arr)
        self._recent_order.append(id(arr))
        if len(self._recent_order) > 256:
            self._recent.pop(self._recent_order.pop(0), None)
        return slot

    def _release(self, slot):
        self._ref[slot] -= 1
        if self._ref[slot] == 0:
            for key in [k for k, v in self._recent.items() if v[0] == slot]:
                del self._recent[key]
            self._free.append(slot)

    def push(self, state, action, reward, next_state):
        if len(self.buffer) < self.capacity:
            self.buffer.append(None)
        old = self.buffer[self.position]
        s_slot = self._upload(state)
        self._ref[s_slot] += 1
        n_slot = None
        if next_state is not None:
            n_slot = self._upload(next_state)
            self._ref[n_slot] += 1
        if old is not None:                                  # the ring wrapped: the overwritten transition lets go
            self._release(int(old.state))
            if old.next_state is not None:
                self._release(int(old.next_state))
        self.buffer[self.position] = Transition(_DeviceObs(self.pool, s_slot), int(action), float(reward),
                                                None if n_slot is None else _DeviceObs(self.pool, n_slot))
        self.position = (self.position + 1) % self.capacity

    def push_many(self, states, actions, rewards, next_states, terminal):
        for i in range(len(actions)):
            self.push(states[i], actions[i], rewards[i], None if terminal[i] else next_states[i])

    @property
    def observations_resident(self):
        return len(self._ref) - len(self._free)


class _HardwareQueues:
    """Which streams of one device share a HARDWARE queue.  The HIP runtime multiplexes a process's streams onto a few hardware queues
    (4 by default), kernels of one hardware queue run in order, and which queue a stream lands on follows the order in which the
    process's streams were first used.  Measured (round 6, tools/leg_order_probe.py): whenever a learner's side stream shared the launch
    stream's queue, its step lost the forward overlap -- 4 090 instead of 4 720 tr/s on configs[1], the whole "third leg of bench.py reads
    13 % low" effect.  So the streams of a step are not taken blindly: two spin kernels (torch.cuda._sleep) on two streams take twice as long
    when the streams share a queue, which sorts candidate streams into classes; a learner gets its side / third / early streams from
    classes other than its launch stream's and, as far as the queues go, from three different ones.  Streams go back to the pool when their
    learner is collected (the suite creates hundreds of nets: a handful of calibrations per process, ~3 ms each)."""
    CYCLES = 1_500_000          # ~0.7 ms per spin kernel

    def __init__(self, device):
        self.device = device
        self.reps = []            # one representative stream per class
        self.free = []            # [(class, stream)] handed back by collected learners
        self.known = {}           # cuda_stream handle -> class
        self.ok = hasattr(torch.cuda, '_sleep')

    def _shared(self, a, b):
        import time
        def both(x, y):
            torch.cuda.synchronize(self.device)
            t = time.perf_counter()
            with torch.cuda.stream(x):
                torch.cuda._sleep(self.CYCLES)
            with torch.cuda.stream(y):
                torch.cuda._sleep(self.CYCLES)
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t
        serial = min(both(a, a), both(a, a))
        return min(both(a, b), both(a, b)) > 0.75 * serial

    def classify(self, stream):
        key = stream.cuda_stream
        c = self.known.get(key)
        if c is None:
            for c, rep in enumerate(self.reps):
                if self._shared(rep, stream):
                    break
            else:
                self.reps.append(stream)
                c = len(self.reps) - 1
            self.known[key] = c
        return c

    def acquire(self, avoid, n):
        """n streams whose classes differ from `avoid` and, as far as possible, from each other."""
        out, used = [], set(avoid)
        if not self.ok:
            return [torch.cuda.Stream(self.device) for _ in range(n)]
        for distinct in (True, False):            # second pass: classes may repeat among the n (fewer queues than roles), still never `avoid`
            tries = 0
            while len(out) < n and tries < 10:
                pick = next((i for i, (c, _) in enumerate(self.free) if c not in (used if distinct else set(avoid))), None)
                if pick is not None:
                    c, st = self.free.pop(pick)
                else:
                    tries += 1
                    st = torch.cuda.Stream(self.device)
                    c = self.classify(st)
                    if c in (used if distinct else set(avoid)):
                        self.free.append((c, st))
                        continue
                out.append(st)
                used.add(c)
        while len(out) < n:                       # (one hardware queue in all: nothing to choose)
            out.append(torch.cuda.Stream(self.device))
        return out

    def acquire_in(self, want, avoid):
        """One stream of class `want` (created and classified until one lands there; any class outside `avoid` after 12 tries)."""
        pick = next((i for i, (c, _) in enumerate(self.free) if c == want), None)
        if pick is not None:
            return self.free.pop(pick)[1]
        for _ in range(12):
            st = torch.cuda.Stream(self.device)
            c = self.classify(st)
            if c == want:
                return st
            self.free.append((c, st))
        return self.acquire(avoid, 1)[0]

    def release(self, streams):
        for st in streams:
            self.free.append((self.classify(st), st))


_HWQ = {}


def _hardware_queues(device):
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    q = _HWQ.get(device)
    if q is None:
        q = _HWQ[device] = _HardwareQueues(device)
    return q


class LearnerStreams:
    """The streams ONE learner (a policy net with its target net, optimiser state and ring: one robot group of train.py:180-195, or an
    intention net) issues its steps on.  Round 6: they belong to the learner, not to the device -- the robot groups of train.py:255-257 and
    the intention nets of train.py:259-261 are independent networks, and with a launch stream (and side / early streams) each their steps run
    side by side on the device instead of one after the other (tests/test_gpu_overlap.py: bit-identical per net to the sequential order).
      launch   None: the step is issued on the caller's current stream (a single learner).  A stream: the step is issued THERE; it is
               ordered behind the caller's stream at the moment of the call, and whoever touches the nets next on another stream is
               ordered behind the step (FCN._order_behind_last_step) -- no join at the end of the loop pass is needed.
      side     target-net forward / weight gradients beside the launch stream (StepOptions.overlap_target_forward)
      third    the policy's no-grad forward of the three-forward form (simq_train_args.third_stream)
      early    the target-net forward that does not wait for the previous step (StepOptions.early_target_forward)
    side / third / early are chosen against the stream the step is issued on (bind): streams TESTED not to share its hardware queue
    (_HardwareQueues)."""
    __slots__ = ('device', 'launch', '_bound', '_side', '_third', '_early', '__weakref__')

    def __init__(self, device, own_launch_stream=False):
        self.device = torch.device(device)
        self.launch = torch.cuda.Stream(self.device) if own_launch_stream else None
        self._bound = self._side = self._third = self._early = None

    def bind(self, main):
        """Choose side / third / early for steps issued on `main` (once per launch stream)."""
        if self._bound == main.cuda_stream:
            return self
        q = _hardware_queues(self.device)
        if self._side is not None:
            q.release([self._side, self._third, self._early])
        avoid = {q.classify(main)} if q.ok else set()
        up = _UPLOAD_STREAMS.get(self.device if self.device.index is not None else torch.device('cuda', torch.cuda.current_device()))
        up_class = q.classify(up) if (q.ok and up is not None) else None
        if up_class is not None and up_class not in avoid:
            # four roles, four hardware queues: launch | side | third | early.  The upload stream (index copies, ring gathers: microseconds)
            # has to share one of them -- the early stream's, whose work waits for those gathers anyway; never side's or third's, where a
            # gather would queue behind a whole forward pass
            self._side, self._third = q.acquire(avoid | {up_class}, 2)
            self._early = q.acquire_in(up_class, avoid)
        else:
            self._side, self._third, self._early = q.acquire(avoid, 3)
        self._bound = main.cuda_stream
        return self

    def _need(self):
        if self._side is None:
            self.bind(self.launch if self.launch is not None else torch.cuda.current_stream(self.device))

    @property
    def side(self):
        self._need()
        return self._side

    @property
    def third(self):
        self._need()
        return self._third

    @property
    def early(self):
        self._need()
        return self._early

    def give_back(self):
        if self._side is not None:
            _hardware_queues(self.device).release([self._side, self._third, self._early])
            self._bound = self._side = self._third = self._early = None


def learner_streams(policy_net, own_launch_stream=None):
    """The LearnerStreams of `policy_net` (created on first use).  own_launch_stream=True gives the learner a launch stream of its own
    (train_groups / bench.py's multi-net workloads do that for every robot group); None leaves it as it is."""
    import weakref
    ls = getattr(policy_net, '_learner_streams', None)
    if ls is None:
        ls = policy_net._learner_streams = LearnerStreams(policy_net.device_, bool(own_launch_stream))
        weakref.finalize(policy_net, _give_back_streams, weakref.ref(ls))
    elif own_launch_stream and ls.launch is None:
        ls.launch = torch.cuda.Stream(ls.device)
    elif own_launch_stream is False:
        ls.launch = None
    return ls


def _give_back_streams(ref):
    ls = ref()
    if ls is not None:
        try:
            ls.give_back()
        except Exception:            # noqa: BLE001  (interpreter shutdown)
            pass


class _OptState:
    """Flat momentum buffer aliased into a torch.optim.SGD's per-parameter state so that
    optimizer.state_dict() / load_state_dict() (train.py:204,331) keep working."""

    def __init__(self, net):
        self.momentum = torch.zeros_like(net.flat_params)
        self.scratch = torch.zeros(4, dtype=torch.float64, device=net.flat_params.device)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=net.flat_params.device)
        self.initialised = False
        self.views = None


def _opt_state(net, optimizer):
    st = getattr(net, '_simq_opt_state', None)
    if st is None:
        st = _OptState(net)
        net._simq_opt_state = st
    if optimizer is None:
        return st
    if st.views is None:
        st.views = net.reference_views(st.momentum)     # logical OIHW shapes, like the parameters (fcn._reference_view)
    params = [getattr(net, pname) for _, pname, _ in net._param_names]
    # adopt momentum buffers that were loaded from a checkpoint (optimizer.load_state_dict, train.py:204): they index like the
    # reference's OIHW tensors -- whether the reference or this package wrote the file -- and copy_ moves them element by
    # logical index into the OHWI storage
    for (name, _, _), p, v in zip(net._param_names, params, st.views):
        buf = optimizer.state.get(p, {}).get('momentum_buffer')
        if buf is not None and (buf.data_ptr() != v.data_ptr() or buf.stride() != v.stride()):
            if tuple(buf.shape) != tuple(v.shape):
                # checkpoints written before momentum buffers were presented in the reference's OIHW shape stored the OHWI
                # storage shape: accept them (permute to the logical shape); anything else is an error
                if buf.dim() == 4 and tuple(buf.permute(0, 3, 1, 2).shape) == tuple(v.shape):
                    buf = buf.permute(0, 3, 1, 2)
                else:
                    raise SimqError('optimizer state of %s has shape %s, expected %s (the reference layout)'
                                    % (name, tuple(buf.shape), tuple(v.shape)))
            v.copy_(buf.to(v.device))
            optimizer.state[p]['momentum_buffer'] = v
            st.initialised = True
    return st


def _hyper(optimizer):
    if optimizer is None or not hasattr(optimizer, 'param_groups'):
        raise SimqError('train: optimizer must expose param_groups (torch.optim.SGD as built at train.py:186)')
    if len(optimizer.param_groups) != 1:
        raise SimqError('train: exactly one parameter group expected (train.py:186)')
    g = optimizer.param_groups[0]
    if g.get('nesterov') or g.get('dampening', 0) != 0 or g.get('maximize'):
        raise SimqError('train: only plain momentum SGD (train.py:186) is implemented')
    return float(g['lr']), float(g.get('momentum', 0.0)), float(g.get('weight_decay', 0.0))


class PendingStep:
    """What train_step(sync='defer') returns: the step is enqueued, `result()` waits for the copy of its four loss sums (issued right behind
    the TD / Huber launch, long before backward + SGD finish) and returns train()'s dict.  Lets a caller enqueue the steps of several
    learners before it waits for any of them (train_groups)."""
    __slots__ = ('_plan', '_host', '_gB', '_info')

    def __init__(self, plan, host, gB):
        self._plan, self._host, self._gB, self._info = plan, host, gB, None

    def result(self):
        if self._info is None:
            lib.call('simq_train_loss_wait', self._plan.handle)
            o = self._host.tolist()                                             # train.py:138-139 (.item())
            self._info = {'td_error': o[1] / self._gB, 'loss': o[0] / self._gB}
        return self._info


def train_step(policy_net, target_net, batch, discount_factor, batch_size, lr, momentum, weight_decay,
               grad_norm_clipping, use_double_dqn=True, opt_state=None, process_group=None, global_batch=None,
               sync=True, comm=None, sync_bn=False, global_nonfinal=None, options=None):
    """One TD step (train.py:108-141) entirely on the device, over the nets' flat buffers.

    options: a StepOptions (default: policy_net.step_options, else DEFAULT_STEP_OPTIONS) -- how this learner issues the step.
    The step is issued on the learner's launch stream when it has one (learner_streams(policy_net, own_launch_stream=True):
    concurrent robot groups), otherwise on the caller's current stream.
    sync: True -- returns {'td_error','loss'} floats (as the reference's .item() calls do); False -- the device tensor [sum_huber, sum_td];
    'defer' (single-process fused step only) -- a PendingStep whose result() gives the dict.

    Data-parallel: pass `global_batch` and either `comm` (simq.dist.Comm: libsimq's RCCL communicator -- the step stays ONE
    library call, the gradient buckets travel on the communicator's stream) or `process_group` (torch.distributed
    collectives around the two backward phases); this rank's `batch` is its slice of the minibatch, BatchNorm uses per-rank
    statistics (the reference's DataParallel semantics, policies.py:39) and the flat gradient is summed over the ranks in two
    buckets before every rank applies the identical clip + SGD.
    sync_bn=True (data-parallel only): the SyncBN option -- the grad-mode forward / backward normalise with the statistics of the
    GLOBAL minibatch (44 small all-reduces per step), which makes the N-rank step the single-device step on the whole minibatch
    instead of DataParallel's per-replica BatchNorm; `global_nonfinal` = number of non-final next states in the WHOLE minibatch
    (every rank draws the same minibatch, so each knows it) sizes the double-DQN forward's global statistics.
    """
    if not isinstance(policy_net, FCN) or not isinstance(target_net, FCN):
        raise SimqError('simq.train needs simq.FCN networks (got %s / %s); there is no torch fallback'
                        % (type(policy_net).__name__, type(target_net).__name__))
    dev = policy_net.device_
    opts = options if options is not None else (getattr(policy_net, 'step_options', None) or DEFAULT_STEP_OPTIONS)
    ls = learner_streams(policy_net)
    caller = torch.cuda.current_stream(dev)
    if ls.launch is not None and ls.launch != caller:
        # this learner's own launch stream: ordered behind the caller's stream as of now (the minibatch, a target sync ... were issued
        # there), then everything below runs with it as the current stream
        ls.launch.wait_stream(caller)
        if isinstance(batch, DeviceBatch):
            for t in (batch.state, batch.action, batch.reward, batch.next_state, batch.nonfinal_pos):
                t.record_stream(ls.launch)
        with torch.cuda.stream(ls.launch):
            return train_step(policy_net, target_net, batch, discount_factor, batch_size, lr, momentum, weight_decay, grad_norm_clipping,
                              use_double_dqn, opt_state, process_group, global_batch, sync, comm, sync_bn, global_nonfinal, opts)
    policy_net._order_behind_last_step()
    target_net._order_behind_last_step()
    parallel = process_group is not None or comm is not None
    if sync_bn and not parallel:
        raise SimqError('train_step: sync_bn needs a process group or a communicator (one process has nothing to synchronise)')
    if sync_bn and global_nonfinal is None:
        raise SimqError('train_step: sync_bn needs global_nonfinal (non-final next states of the whole minibatch)')
    b = assemble_batch(batch, dev, allow_all_final=parallel)
    B = b.state.shape[0]
    if b.next_state.shape[0] == 0 and not parallel:
        raise SimqError('train: batch has no non-final next state (the reference raises at train.py:112 too)')
    if B != batch_size:
        raise SimqError('train: batch has %d transitions but cfg.batch_size is %d' % (B, batch_size))
    gB = B if global_batch is None else int(global_batch)
    st = stream_ptr(dev)
    n = policy_net.num_output_channels * W * W
    st_opt = opt_state if opt_state is not None else _opt_state(policy_net, None)
    # the next-state forwards see a different number of non-final samples every step: size their workspaces for the whole
    # batch once, so that no step re-allocates gigabytes when it draws fewer terminal transitions than any step before
    policy_net._workspace('tmp', B)
    target_net._workspace('tmp', B)

    if (process_group is None or comm is not None) and opts.fused:
        return _train_step_fused(policy_net, target_net, b, discount_factor, gB, lr, momentum, weight_decay, grad_norm_clipping,
                                 use_double_dqn, st_opt, sync, comm, sync_bn, global_nonfinal, opts, ls)
    if sync == 'defer':
        raise SimqError("train_step: sync='defer' needs the fused single-process step")
    bn_sync = sdist.SyncBN(gB, group=process_group, comm=comm) if sync_bn else None
    bn_sync_nf = sdist.SyncBN(global_nonfinal, group=process_group, comm=comm) if (sync_bn and global_nonfinal) else None
    reduce_async = (lambda t: sdist.allreduce_async(t, process_group)) if comm is None else comm.all_reduce

    # train.py:114 -- policy forward, train-mode BN, activations kept for backward
    q = policy_net._forward_raw(b.state, MODE_TRAIN, sync=bn_sync)
    # The target-net forward (eval mode, its own parameters / workspace) depends on nothing the policy net computes: it runs
    # on a side stream, forked behind the train-mode forward so that it overlaps the policy's next-state forward (both work
    # on the ~29 non-final samples and fill each other's partially filled rounds of CUs).
    main = torch.cuda.current_stream(dev)
    ls.bind(main)
    side = ls.side if opts.overlap_target_forward else main
    Nn = b.next_state.shape[0]
    nsv = torch.empty(B, dtype=torch.float32, device=dev)
    vals = torch.empty(max(Nn, 1), dtype=torch.float32, device=dev)
    if Nn:           # (Nn == 0: an all-terminal data-parallel shard -- no bootstrap values, but it joins the collectives below)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            q_tgt = target_net._forward_raw(b.next_state, MODE_EVAL)          # train.py:122/124 (target in eval mode)
    # train.py:116-124 -- bootstrap values of the non-final next states
    if Nn == 0:
        if use_double_dqn and bn_sync_nf is not None:       # all-terminal shard under SyncBN: zeros into the others' reductions
            ws0 = policy_net._workspace('tmp', 1)
            lib.call('simq_forward_sync_null', policy_net.plan.handle, 1, ptr(policy_net.bn_buffers), ptr(ws0), st, bn_sync_nf.bind(ws0))
            for k in policy_net.num_batches_tracked:            # the global statistics were committed to this rank's buffers too
                policy_net.num_batches_tracked[k] += 1
    elif use_double_dqn:
        # train.py:121: the POLICY net, still in train mode (batch statistics, 2nd running-stat update)
        q_next = policy_net._forward_raw(b.next_state, MODE_TRAIN_NOGRAD, sync=bn_sync_nf)
        best = torch.empty(Nn, dtype=torch.int64, device=dev)
        lib.call('simq_q_argmax', ptr(q_next), Nn, n, ptr(best), None, st)
        main.wait_stream(side)
        lib.call('simq_q_gather', ptr(q_tgt), Nn, n, ptr(best), ptr(vals), st)
    else:
        main.wait_stream(side)
        lib.call('simq_q_argmax', ptr(q_tgt), Nn, n, None, ptr(vals), st)
    if Nn:
        q_tgt.record_stream(main)
    lib.call('simq_scatter_next_values', ptr(vals), ptr(b.nonfinal_pos), Nn, ptr(nsv), B, st)
    # train.py:115,126-129 + the gradient autograd would hand to `output`
    q_sa = torch.empty(B, dtype=torch.float32, device=dev)
    y = torch.empty(B, dtype=torch.float32, device=dev)
    td = torch.empty(B, dtype=torch.float32, device=dev)
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    # (the upstream gradient stays in its one-hot form: B non-zeros, the backward walk starts from those pixels)
    lib.call('simq_td_huber', ptr(q), B, n, ptr(b.action), ptr(b.reward), ptr(nsv), float(discount_factor),
             1.0 / gB, ptr(q_sa), ptr(y), ptr(td), ptr(out4), None, st)
    # train.py:131-132
    if not parallel:
        grads = policy_net._backward_onehot(b.action, q_sa, y, 1.0 / gB, B)
    else:
        # data parallel: the all-reduce of the head + layer4 gradients (75 % of the 45 MB) is issued as soon as they are
        # final and runs on RCCL's stream while layers 3..1 + stem are still being differentiated
        split = policy_net.grad_bucket_split
        grads = policy_net._backward_onehot(b.action, q_sa, y, 1.0 / gB, B, phase=1, sync=bn_sync)
        works = [reduce_async(grads[split:])]
        policy_net._backward_onehot(b.action, q_sa, y, 1.0 / gB, B, phase=2, sync=bn_sync)
        works += [reduce_async(grads[:split]), reduce_async(out4)]
        if comm is not None:
            comm.wait()
        else:
            for wk in works:
                wk.wait()
    # train.py:133-135
    lib.call('simq_clip_sgd_step', ptr(policy_net.flat_params), ptr(grads), ptr(st_opt.momentum),
             policy_net.plan.param_count, float(grad_norm_clipping) if grad_norm_clipping is not None else 0.0,
             lr, momentum, weight_decay, 0 if st_opt.initialised else 1, ptr(st_opt.scratch), ptr(st_opt.total_norm), st)
    st_opt.initialised = True
    policy_net.weights_dirty = True      # parameters moved: the next forward refreshes the derived weight cache
    policy_net._last = {'q_sa': q_sa, 'y': y, 'td': td, 'q': q}
    policy_net._mark_step(main)
    target_net._mark_step(main)
    if not sync:
        return out4
    o = out4.tolist()                                                       # train.py:138-139 (.item() host sync)
    return {'td_error': o[1] / gB, 'loss': o[0] / gB}


def _train_step_fused(policy_net, target_net, b, discount_factor, gB, lr, momentum, weight_decay, grad_norm_clipping,
                      use_double_dqn, st_opt, sync, comm=None, sync_bn=False, global_nonfinal=None, opts=DEFAULT_STEP_OPTIONS, ls=None):
    """train_step through simq_train_step: the same launches in the same order, sequenced inside the library (with `comm`: the
    data-parallel form, gradient buckets all-reduced on the communicator's stream between the backward phases and the SGD)."""
    dev = policy_net.device_
    B, Nn = b.state.shape[0], b.next_state.shape[0]
    Nn_real, Nn = Nn, max(Nn, 1)          # (all-terminal shard: the scratch tensors keep one row, the library is told 0)
    n = policy_net.num_output_channels * W * W
    for t, c in ((b.state, policy_net.num_input_channels), (b.next_state, policy_net.num_input_channels)):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape[1:]) != (W, W, c):
            raise SimqError('train: states must be contiguous fp32 [B,%d,%d,%d] (NHWC), got %s' % (W, W, c, tuple(t.shape)))
    policy_net._ensure_weights()
    target_net._ensure_weights()
    f32 = dict(dtype=torch.float32, device=dev)
    q, q_tgt, dq = torch.empty((B, n), **f32), torch.empty((Nn, n), **f32), None     # dq None: one-hot backward
    q_next = torch.empty((Nn, n), **f32) if use_double_dqn else None
    best = torch.empty(Nn, dtype=torch.int64, device=dev) if use_double_dqn else None
    vec = torch.empty(5 * B + 4, **f32)
    nsv, vals, q_sa, y, td, out4 = vec[:B], vec[B:2 * B], vec[2 * B:3 * B], vec[3 * B:4 * B], vec[4 * B:5 * B], vec[5 * B:]
    main = torch.cuda.current_stream(dev)
    if ls is None:
        ls = learner_streams(policy_net)
    ls.bind(main)
    side = ls.side if opts.overlap_target_forward else None
    a = TrainArgs()
    a.struct_bytes = ctypes.sizeof(TrainArgs)
    a.plan = policy_net.plan.handle
    a.batch, a.num_nonfinal, a.global_batch = B, Nn_real, gB
    a.comm = comm.handle if comm is not None else None
    a.sync_bn = 1 if (sync_bn and comm is not None) else 0
    a.global_nonfinal = int(global_nonfinal) if global_nonfinal is not None else Nn_real
    next_state = b.next_state if Nn_real else torch.empty((1, W, W, policy_net.num_input_channels), **f32)
    nonfinal_pos = b.nonfinal_pos if Nn_real else torch.zeros(1, dtype=torch.int32, device=dev)
    a.use_double_dqn, a.first_step = int(bool(use_double_dqn)), 0 if st_opt.initialised else 1
    a.gamma, a.lr, a.momentum, a.weight_decay = float(discount_factor), lr, momentum, weight_decay
    a.max_norm = float(grad_norm_clipping) if grad_norm_clipping is not None else 0.0
    t_ws = target_net._workspace('tmp', Nn)
    tensors = dict(params=policy_net.flat_params, wcache=policy_net.wcache, bnbuf=policy_net.bn_buffers, grads=policy_net.flat_grads,
                   momentum_buf=st_opt.momentum, ws_train=policy_net._workspace('train', B), ws_tmp=policy_net._workspace('tmp', Nn),
                   t_params=target_net.flat_params, t_wcache=target_net.wcache, t_bnbuf=target_net.bn_buffers,
                   t_ws=t_ws, state=b.state, next_state=next_state, action=b.action, reward=b.reward,
                   nonfinal_pos=nonfinal_pos, q=q, q_next=q_next, q_tgt=q_tgt, dq=dq, nsv=nsv, vals=vals, best=best, q_sa=q_sa,
                   y=y, td=td, out4=out4, opt_scratch=st_opt.scratch, total_norm=st_opt.total_norm)
    # The target net's forward over the next states (train.py:122) reads nothing this step or the previous one computes.  When the
    # minibatch was gathered on a stream of its own (DeviceBatch.ready_event) it goes to an "early" stream ordered behind exactly what it
    # reads -- the gathered next states, the target net's weight cache, the last reader of the Q-map buffer it writes -- and runs beside
    # the previous step's backward pass and SGD (the host enqueues this step while the previous one still runs: `sync` below).  Same
    # kernels on the same operands: bit-identical.  The Q-map buffers alternate between two per net (a fresh torch.empty could be a block
    # the previous step's queued kernels still use: the allocator only knows the consuming stream).
    early = None
    if (opts.early_target_forward and side is not None and use_double_dqn and Nn_real and not a.sync_bn and getattr(b, 'ready_event', None) is not None
            and policy_net.plan.options.get('fwd_overlap', 2) == 2):
        early = ls.early
        slot = policy_net.__dict__.setdefault('_qtgt_slot', 0)
        bufs = policy_net.__dict__.setdefault('_qtgt_bufs', [None, None])
        frees = policy_net.__dict__.setdefault('_qtgt_free', [None, None])
        fresh = False
        if bufs[slot] is None or bufs[slot].numel() < Nn * n:
            bufs[slot] = torch.empty(Nn * n, **f32)
            frees[slot] = None
            fresh = True
        # a buffer the early stream writes that was allocated just now (the Q-map slot, or the target net's forward workspace after it grew)
        # comes from the launch stream's pool and may be a block kernels still queued THERE are using: the early stream waits for them once
        # (in steady state nothing is allocated and the early stream is ordered behind its inputs only)
        if fresh or policy_net.__dict__.get('_early_tws') != t_ws.data_ptr():
            early.wait_stream(main)
            policy_net._early_tws = t_ws.data_ptr()
        q_tgt = bufs[slot][:Nn * n].view(Nn, n)
        tensors['q_tgt'] = q_tgt
        early.wait_event(b.ready_event)
        if getattr(target_net, '_weights_event', None) is not None:
            early.wait_event(target_net._weights_event)
        if frees[slot] is not None:
            early.wait_event(frees[slot])
        b.next_state.record_stream(early)
        policy_net._qtgt_slot = slot ^ 1
    for k, t in tensors.items():
        setattr(a, k, None if t is None else t.data_ptr())
    a.stream = main.cuda_stream
    a.side_stream = side.cuda_stream if side is not None else None
    a.target_stream = early.cuda_stream if early is not None else None
    a.third_stream = ls.third.cuda_stream if side is not None else None
    if comm is not None and side is not None:
        # the gradient buckets travel on the learner's third stream: idle during the backward pass, and TESTED not to share the hardware queue of
        # the launch stream (dgrads) or of the side stream (weight gradients) -- the communicator's own stream lands wherever the runtime puts it
        comm.adopt_stream(ls.third)
    loss_host = None
    if sync:
        # train.py:137-139 (loss.item()) without synchronising the stream: the library copies the four sums to pinned memory as soon as
        # they are final (behind the TD / Huber launch) and the host waits for THAT copy -- the next step is enqueued while this one runs
        loss_host = getattr(policy_net, '_loss_host', None)
        if loss_host is None:
            loss_host = policy_net._loss_host = torch.empty(4, dtype=torch.float32).pin_memory()
        a.loss_host = loss_host.data_ptr()
    lib.call('simq_train_step', ctypes.byref(a))
    if early is not None:
        ev = torch.cuda.Event()
        ev.record(main)                     # (behind this step's q_gather: the buffer may be rewritten two steps from now)
        policy_net._qtgt_free[policy_net._qtgt_slot ^ 1] = ev
    elif side is not None:
        q_tgt.record_stream(side)
    # bookkeeping the separate calls do on the Python side
    policy_net._train_generation += 1
    for k in policy_net.num_batches_tracked:
        # (an all-terminal shard under SyncBN still commits the second, global, running-stat update: simq_forward_sync_null)
        second = use_double_dqn and (Nn_real or (a.sync_bn and a.global_nonfinal > 0))
        policy_net.num_batches_tracked[k] += 2 if second else 1
    policy_net.weights_dirty = False          # the library refreshed the weight cache behind the SGD update
    policy_net._weights_stamp += 1
    st_opt.initialised = True
    policy_net._last = {'q_sa': q_sa, 'y': y, 'td': td, 'q': q.view(B, policy_net.num_output_channels, W, W)}
    # simq_train_step convolved the minibatch in place: the workspace holds no copy of it, so a backward entry point called on its own
    # before the next grad-mode forward would differentiate the stem against a stale input (FCN._backward_* refuse)
    policy_net._train_input_inplace = True
    policy_net._mark_step(main)
    target_net._mark_step(main)
    if not sync:
        return out4
    pending = PendingStep(policy_net.plan, loss_host, gB)
    return pending if sync == 'defer' else pending.result()


def train_step_dataparallel(policy_net, target_net, global_batch, discount_factor, lr, momentum, weight_decay, grad_norm_clipping,
                            process_group, opt_state=None, sync=True):
    """One double-DQN TD step as nn.DataParallel LITERALLY runs it on N devices (policies.py:39 around train.py:108-141), one process per
    device.  `global_batch`: the WHOLE sampled minibatch (every rank holds the ring and draws the same indices).  Rank r
      * forwards rows shard_bounds(B, N, r) of the states (train-mode BatchNorm over its rows, as in train_step), and
      * picks the greedy next actions of torch.chunk piece r of the COMPACTED non-final next states -- DataParallel scatters the tensor
        train.py:121 hands it, so those rows belong to whichever transitions they came from, not to this rank's slice (the default
        `train_step` sharding keeps next states with their transitions: simpler, no exchange, same semantics class) --;
    the pieces are exchanged (simq.dist.gather_greedy_actions), the target net (eval mode: chunking is immaterial) evaluates the next
    states of THIS rank's transitions, and loss / backward / the two-bucket gradient all-reduce / clip + SGD follow as in train_step.
    Rank 0's running statistics see state chunk 0 and compacted chunk 0, exactly replica 0's.  Fixture: tests/golden/dplit_*.npz
    (oracle/gen_golden.py dp_literal, the reference's own modules replica by replica)."""
    if process_group is None:
        raise SimqError('train_step_dataparallel needs a torch.distributed process group')
    import torch.distributed as tdist
    world, rank = tdist.get_world_size(process_group), tdist.get_rank(process_group)
    dev = policy_net.device_
    g = assemble_batch(global_batch, dev)
    gB = g.state.shape[0]
    # every rank holds the same minibatch, so every rank reaches the same verdict BEFORE any collective: torch.chunk (DataParallel's
    # scatter) hands out ceil(gB / world) rows per replica, and with gB <= chunk * (world - 1) the last replicas get none -- DataParallel
    # would simply use fewer devices; one process per device cannot drop out of the all-reduces, so this is an error on ALL ranks
    chunk = -(-gB // world)
    if chunk * (world - 1) >= gB:
        raise SimqError('train_step_dataparallel: a %d-transition minibatch in chunks of %d leaves rank(s) from %d on (of %d) without rows'
                        % (gB, chunk, -(-gB // chunk), world))
    lo, hi = sdist.shard_bounds(gB, world, rank)
    B = hi - lo
    st = stream_ptr(dev)
    n = policy_net.num_output_channels * W * W
    st_opt = opt_state if opt_state is not None else _opt_state(policy_net, None)
    state = g.state[lo:hi]
    q = policy_net._forward_raw(state, MODE_TRAIN)                                                   # train.py:114, rows of replica r
    Nn = g.next_state.shape[0]
    clo, chi = sdist.shard_bounds(Nn, world, rank)                                                   # torch.chunk piece r of the compacted tensor
    best_chunk = torch.empty(chi - clo, dtype=torch.int64, device=dev)
    if chi > clo:                                                                                    # train.py:121 on replica r
        q_next = policy_net._forward_raw(g.next_state[clo:chi], MODE_TRAIN_NOGRAD)
        lib.call('simq_q_argmax', ptr(q_next), chi - clo, n, ptr(best_chunk), None, st)
    best = sdist.gather_greedy_actions(best_chunk, Nn, world, rank, process_group)
    # the next states of THIS rank's transitions: compacted indices k0 .. k1-1 (positions are increasing)
    pos = [i for i, m in enumerate(g.non_final_mask) if m]
    k0 = sum(1 for p in pos if p < lo)
    k1 = sum(1 for p in pos if p < hi)
    nsv = torch.empty(B, dtype=torch.float32, device=dev)
    vals = torch.empty(max(k1 - k0, 1), dtype=torch.float32, device=dev)
    own_pos, = _upload_packed(dev, [np.asarray([p - lo for p in pos[k0:k1]] or [0], dtype=np.int32)])   # (pinned, asynchronous: no stream stall)
    if k1 > k0:
        q_tgt = target_net._forward_raw(g.next_state[k0:k1], MODE_EVAL)                              # train.py:122
        lib.call('simq_q_gather', ptr(q_tgt), k1 - k0, n, ptr(best[k0:k1].contiguous()), ptr(vals), st)
    lib.call('simq_scatter_next_values', ptr(vals), ptr(own_pos), k1 - k0, ptr(nsv), B, st)
    action, reward = g.action[lo:hi].contiguous(), g.reward[lo:hi].contiguous()
    q_sa = torch.empty(B, dtype=torch.float32, device=dev)
    y = torch.empty(B, dtype=torch.float32, device=dev)
    td = torch.empty(B, dtype=torch.float32, device=dev)
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    lib.call('simq_td_huber', ptr(q), B, n, ptr(action), ptr(reward), ptr(nsv), float(discount_factor), 1.0 / gB, ptr(q_sa), ptr(y),
             ptr(td), ptr(out4), None, st)
    split = policy_net.grad_bucket_split
    grads = policy_net._backward_onehot(action, q_sa, y, 1.0 / gB, B, phase=1)
    works = [sdist.allreduce_async(grads[split:], process_group)]
    policy_net._backward_onehot(action, q_sa, y, 1.0 / gB, B, phase=2)
    works += [sdist.allreduce_async(grads[:split], process_group), sdist.allreduce_async(out4, process_group)]
    for wk in works:
        wk.wait()
    lib.call('simq_clip_sgd_step', ptr(policy_net.flat_params), ptr(grads), ptr(st_opt.momentum), policy_net.plan.param_count,
             float(grad_norm_clipping) if grad_norm_clipping is not None else 0.0, lr, momentum, weight_decay,
             0 if st_opt.initialised else 1, ptr(st_opt.scratch), ptr(st_opt.total_norm), st)
    st_opt.initialised = True
    policy_net.weights_dirty = True
    policy_net._last = {'q_sa': q_sa, 'y': y, 'td': td, 'q': q, 'best': best}
    if not sync:
        return out4
    o = out4.tolist()
    return {'td_error': o[1] / gB, 'loss': o[0] / gB}


def train(cfg, policy_net, target_net, optimizer, batch, transform_fn, discount_factor):
    """Drop-in for train.train (train.py:108-141).  `transform_fn` (policies.py:44-45) is
    accepted for signature compatibility; the HIP path reads the HWC states as they are."""
    lr, momentum, weight_decay = _hyper(optimizer)
    st = _opt_state(policy_net, optimizer)
    info = train_step(policy_net, target_net, batch, discount_factor, cfg.batch_size, lr, momentum, weight_decay,
                      cfg.grad_norm_clipping, use_double_dqn=cfg.use_double_dqn, opt_state=st)
    _expose_momentum(policy_net, optimizer, st, momentum)
    return info


def train_intention_step(intention_net, batch, lr, momentum, weight_decay, opt_state=None, process_group=None,
                         global_batch=None, sync=True, comm=None, _on_launch_stream=False):
    """One intention-map supervision step (train.py:143-158) on the device: split the ground-truth map (last state
    channel) off the replay states, train-mode forward of FCN(C-1, 1), BCE-with-logits + its gradient in one kernel,
    backward, momentum SGD (no clipping on this path).  Returns {'loss_intention': float} or, with sync=False, the
    device double holding the summed loss."""
    if not isinstance(intention_net, FCN):
        raise SimqError('simq.train_intention needs a simq.FCN network (got %s); there is no torch fallback'
                        % type(intention_net).__name__)
    if intention_net.num_output_channels != 1:
        raise SimqError('train_intention: the intention net has one output channel (policies.py:93)')
    dev = intention_net.device_
    ls = learner_streams(intention_net)
    caller = torch.cuda.current_stream(dev)
    if ls.launch is not None and ls.launch != caller and not _on_launch_stream:
        # the intention net's own launch stream (train_intention_groups: beside the robot groups' TD steps)
        ls.launch.wait_stream(caller)
        if isinstance(batch, DeviceBatch):
            batch.state.record_stream(ls.launch)
        with torch.cuda.stream(ls.launch):
            return train_intention_step(intention_net, batch, lr, momentum, weight_decay, opt_state, process_group, global_batch, sync, comm, True)
    intention_net._order_behind_last_step()
    if isinstance(batch, DeviceBatch):
        full = batch.state
    else:
        full = torch.from_numpy(np.stack(batch.state)).to(dev, non_blocking=True)
    B, C = full.shape[0], full.shape[3]
    if C != intention_net.num_input_channels + 1:
        raise SimqError('train_intention: states have %d channels, expected %d (+1 ground-truth intention map)'
                        % (C, intention_net.num_input_channels))
    gB = B if global_batch is None else int(global_batch)
    st = stream_ptr(dev)
    st_opt = opt_state if opt_state is not None else _opt_state(intention_net, None)
    x = torch.empty((B, W, W, C - 1), dtype=torch.float32, device=dev)
    target = torch.empty((B, W, W), dtype=torch.float32, device=dev)
    lib.call('simq_split_last_channel', ptr(full), ptr(x), ptr(target), B * W * W, C, st)          # train.py:145-146
    logits = intention_net._forward_raw(x, MODE_TRAIN)                                                # train.py:148
    dlogits = torch.empty_like(logits)
    loss_sum = torch.empty(1, dtype=torch.float64, device=dev)
    lib.call('simq_bce_with_logits', ptr(logits), ptr(target), B * W * W, ptr(dlogits), ptr(loss_sum), st)   # :149-150
    if gB != B:
        dlogits.mul_(B / gB)
    if process_group is None and comm is None:
        grads = intention_net._backward_raw(dlogits, B)                                               # train.py:151-152
    else:       # data parallel: same two gradient buckets as train_step
        reduce_async = (lambda t: sdist.allreduce_async(t, process_group)) if comm is None else comm.all_reduce
        split = intention_net.grad_bucket_split
        grads = intention_net._backward_raw(dlogits, B, phase=1)
        works = [reduce_async(grads[split:])]
        intention_net._backward_raw(dlogits, B, phase=2)
        works += [reduce_async(grads[:split]), reduce_async(loss_sum)]
        if comm is not None:
            comm.wait()
        else:
            for wk in works:
                wk.wait()
    lib.call('simq_clip_sgd_step', ptr(intention_net.flat_params), ptr(grads), ptr(st_opt.momentum),
             intention_net.plan.param_count, 0.0, lr, momentum, weight_decay, 0 if st_opt.initialised else 1,
             ptr(st_opt.scratch), ptr(st_opt.total_norm), st)                                         # train.py:153
    st_opt.initialised = True
    intention_net.weights_dirty = True
    intention_net._last = {'logits': logits, 'target': target}
    intention_net._mark_step(torch.cuda.current_stream(dev))
    if not sync:
        return loss_sum
    if sync == 'defer':
        return _PendingIntention(loss_sum, gB, torch.cuda.current_stream(dev))
    return {'loss_intention': float(loss_sum.item()) / (gB * W * W)}                                  # train.py:155-156


class _PendingIntention:
    """train_intention_step(sync='defer'): the summed loss is read back on the stream that produced it when result() is called."""
    __slots__ = ('_loss', '_gB', '_stream', '_info')

    def __init__(self, loss_sum, gB, stream):
        self._loss, self._gB, self._stream, self._info = loss_sum, gB, stream, None

    def result(self):
        if self._info is None:
            with torch.cuda.stream(self._stream):
                self._info = {'loss_intention': float(self._loss.item()) / (self._gB * W * W)}       # train.py:155-156
        return self._info


def train_intention(intention_net, optimizer, batch, transform_fn):
    """Drop-in for train.train_intention (train.py:143-158); `transform_fn` kept for signature compatibility."""
    lr, momentum, weight_decay = _hyper(optimizer)
    st = _opt_state(intention_net, optimizer)
    info = train_intention_step(intention_net, batch, lr, momentum, weight_decay, opt_state=st)
    _expose_momentum(intention_net, optimizer, st, momentum)
    return info


def _expose_momentum(net, optimizer, st, momentum):
    if momentum != 0:   # the (aliased) momentum buffers exactly where torch.optim.SGD keeps them
        params = [getattr(net, pname) for _, pname, _ in net._param_names]
        for p, v in zip(params, st.views):
            optimizer.state[p]['momentum_buffer'] = v


def train_groups(cfg, policy_nets, target_nets, optimizers, batches, transform_fn, discount_factors,
                 intention_nets=None, optimizers_intention=None, concurrent=False):
    """One pass of the reference's training loop body over ALL robot groups (train.py:253-261) as one call:

        for i in range(num_robot_groups):
            train_info = train(cfg, policy_nets[i], target_nets[i], optimizers[i], batches[i], transform_fn, discount_factors[i])
            if cfg.use_predicted_intention:
                train_info.update(train_intention(intention_nets[i], optimizers_intention[i], batches[i], transform_fn))

    The groups' Q-networks and the intention networks are independent networks with their own parameters, optimiser state, workspaces
    and plans.  Every step is enqueued before the first loss is waited for.  With concurrent=True every learner also issues its step on a launch
    stream of its own (LearnerStreams): measured on one MI355X this is level with the sequential order (configs[3]: 5 362 against 5 349 tr/s at 256
    per net, 5 009 against 5 047 at 64 -- a single learner's four streams already occupy the four hardware queues and never more than two
    kernels fit the device at once), hence off by default; the mechanism is tested bit-identical and is there for parts / runtimes where it pays.
    Per net the same kernels run on the same operands in the same order either way: every net's results are what the sequential loop gives
    (bit-identical on deterministic plans, tests/test_gpu_overlap.py).
    No join at the end: anything that touches one of the nets later on another stream is ordered behind its step
    (FCN._order_behind_last_step).
    batches[i]: what replay_buffers[i].sample(cfg.batch_size) returned (a DeviceBatch or the reference's Transition-of-tuples), or None to
    skip the group.  Returns the list of train_info dicts (None for skipped groups)."""
    n = len(policy_nets)
    pend, pend_int = [None] * n, [None] * n
    for i in range(n):
        if batches[i] is None:
            continue
        lr, momentum, weight_decay = _hyper(optimizers[i])
        st = _opt_state(policy_nets[i], optimizers[i])
        learner_streams(policy_nets[i], own_launch_stream=True if concurrent else None)
        pend[i] = train_step(policy_nets[i], target_nets[i], batches[i], discount_factors[i], cfg.batch_size, lr, momentum, weight_decay,
                             cfg.grad_norm_clipping, use_double_dqn=cfg.use_double_dqn, opt_state=st, sync='defer')
        _expose_momentum(policy_nets[i], optimizers[i], st, momentum)
        if intention_nets is not None:
            lr, momentum, weight_decay = _hyper(optimizers_intention[i])
            st = _opt_state(intention_nets[i], optimizers_intention[i])
            learner_streams(intention_nets[i], own_launch_stream=True if concurrent else None)
            pend_int[i] = train_intention_step(intention_nets[i], batches[i], lr, momentum, weight_decay, opt_state=st, sync='defer')
            _expose_momentum(intention_nets[i], optimizers_intention[i], st, momentum)
    infos = [None] * n
    for i in range(n):
        if pend[i] is not None:
            infos[i] = dict(pend[i].result())
            if pend_int[i] is not None:
                infos[i].update(pend_int[i].result())
    return infos
