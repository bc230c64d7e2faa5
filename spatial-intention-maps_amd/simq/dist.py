"""Data-parallel plumbing for the TD step (one process per GPU, RCCL over xGMI).

The reference scales with single-process nn.DataParallel (policies.py:39): the minibatch is
scattered over the GPUs, every replica normalises with ITS OWN BatchNorm batch statistics, the
gradients are reduce-added onto device 0 and only device 0's running statistics persist.  The
MI355X-native equivalent keeps exactly those semantics with one process per GPU:

  * rank r trains on the contiguous slice shard_bounds(global_batch, world, r) of the sampled
    minibatch (every rank holds the whole replay ring and draws the same indices -> no data exchange),
  * the Huber mean is over the GLOBAL batch (each rank scales its one-hot dLoss/dQ by 1/global_batch),
  * ONE all-reduce(sum) of the flat gradient buffer (11.25 M fp32 = 45 MB) + 4 loss scalars,
  * every rank then applies the identical clip + SGD update (weights stay bit-identical),
  * running statistics are per-rank; rank 0's are the ones that count: broadcast_bn_buffers()
    before a target sync / checkpoint.

Two transports for the gradient sum:
  * `Comm` -- libsimq's own RCCL communicator (include/simq.h simq_comm_*): collectives run on a library-owned HIP stream,
    ordered by events behind the backward kernels; torch.distributed is used ONCE, to hand rank 0's 128-byte RCCL
    identifier to the other ranks.  With it the whole data-parallel step is a single library call (simq_train_step).
  * torch.distributed collectives on the caller's process group ("nccl" == RCCL on ROCm; "gloo" in the CPU tests and when
    several ranks share one GPU, which RCCL refuses).
The sharding helpers below never touch the HIP library, so the rank logic is testable without a GPU.
"""
import ctypes

import torch
import torch.distributed as dist


def shard_bounds(global_batch, world_size, rank):
    """Contiguous [lo, hi) slice of the global minibatch owned by `rank` (torch.chunk sizes,
    like DataParallel's scatter)."""
    if global_batch < 1 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError('shard_bounds: bad arguments %r' % ((global_batch, world_size, rank),))
    chunk = -(-global_batch // world_size)
    lo = min(rank * chunk, global_batch)
    hi = min(lo + chunk, global_batch)
    return lo, hi


def shard_indices(indices, world_size, rank):
    lo, hi = shard_bounds(len(indices), world_size, rank)
    return list(indices[lo:hi])


def gather_greedy_actions(best_chunk, n_total, world_size, rank, group=None):
    """DataParallel's literal scatter of the double-DQN forward (policies.py:39 around train.py:121): rank r picked the greedy actions of
    torch.chunk piece r of the COMPACTED non-final next states (`best_chunk`, int64 on any device, possibly empty); every rank needs the
    actions of the next states of ITS transitions, so the pieces are assembled on all ranks (a sum of disjointly filled vectors --
    n_total int64 values, the only data-path exchange besides the gradient sum).  Returns int64 [n_total] on best_chunk's device."""
    lo, hi = shard_bounds(n_total, world_size, rank) if n_total > 0 else (0, 0)
    full = torch.zeros(max(n_total, 1), dtype=torch.int64, device=best_chunk.device)
    if hi > lo:
        full[lo:hi] = best_chunk
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    return full[:n_total]


def allreduce_gradients(flat_grads, loss_sums=None, group=None):
    """Sum the flat gradient buffer (and the per-rank loss partial sums) over all ranks, in place."""
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    if loss_sums is not None:
        dist.all_reduce(loss_sums, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def allreduce_async(tensor, group=None):
    """Sum `tensor` over ranks in place without blocking the caller's stream; `.wait()` the returned work before the
    result is consumed (NCCL/RCCL: makes the current stream wait for the collective's stream)."""
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)


def broadcast_bn_buffers(bn_buffers, group=None, src=0):
    """DataParallel keeps only device 0's running statistics: make every rank adopt rank `src`'s."""
    dist.broadcast(bn_buffers, src=src, group=group)
    return bn_buffers


def max_over_ranks(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class SyncBN:
    """The SyncBN option (include/simq.h simq_sync): train-mode BatchNorm over the GLOBAL minibatch instead of the
    per-replica statistics of nn.DataParallel.  Wraps the reduction the library calls back for every BatchNorm's partial sums:
    over torch.distributed (`group`: "nccl" == RCCL, or gloo) or over libsimq's own communicator (`comm`), whose native
    simq_comm_reduce_f64 then runs without entering Python."""

    def __init__(self, global_batch, group=None, comm=None):
        from . import _lib
        self.world = comm.world if comm is not None else dist.get_world_size(group)
        self.global_batch = int(global_batch)
        self._group, self._comm, self._ws = group, comm, None
        self.args = _lib.SyncArgs()
        if comm is not None:
            fn = ctypes.cast(_lib._c.simq_comm_reduce_f64, _lib.REDUCE_FN)
            self.args.reduce, self.args.user = fn, comm.handle
        else:
            self._cb = _lib.REDUCE_FN(self._reduce)          # keep the callback object alive
            self.args.reduce, self.args.user = self._cb, None
        self.args.global_batch, self.args.world_size = self.global_batch, self.world

    def bind(self, workspace):
        """The partial sums live inside `workspace` (a torch uint8 tensor): the callback needs it to wrap the raw pointer."""
        self._ws = workspace
        return ctypes.byref(self.args)

    def _reduce(self, user, d_buf, count, stream):
        try:
            off = int(d_buf) - self._ws.data_ptr()
            t = self._ws[off:off + 8 * int(count)].view(torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._group)
            return 0
        except Exception:                                    # noqa: BLE001  (an exception must not cross the C boundary)
            import traceback
            traceback.print_exc()
            return -5


class Comm:
    """libsimq's RCCL communicator (simq_comm_*, include/simq.h) for the ranks of a torch.distributed process group.

    Construction is collective: rank 0 draws the RCCL identifier, torch.distributed broadcasts its 128 bytes (the only use of
    the process group), every rank joins with its CURRENT HIP device.  all_reduce / broadcast are asynchronous with respect to
    the host and to torch's current stream: they are enqueued on the communicator's own stream behind the work already
    submitted to the current stream; wait() makes the current stream wait for them."""

    def __init__(self, group=None, device=None):
        from ._lib import COMM_ID_BYTES, SimqError, lib
        self._lib = lib
        self.handle = None
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        on_device = dist.get_backend(group) == 'nccl'
        where = self.device if on_device else 'cpu'
        # Construction is collective and so is its failure: rank 0's identifier travels with a status byte, and the ranks agree
        # (MIN) on whether every simq_comm_init succeeded -- either all of them hold a communicator or all of them raise, so a
        # caller that falls back to another transport does so on every rank (no rank is left waiting in a collective).
        ident = (ctypes.c_ubyte * COMM_ID_BYTES)()
        status, why = 1, ''
        if self.rank == 0:
            try:
                lib.call('simq_comm_unique_id', ident)
            except Exception as ex:                          # noqa: BLE001  (e.g. librccl cannot be loaded)
                status, why = 0, str(ex)
        t = torch.tensor([status] + list(ident), dtype=torch.uint8, device=where)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        got = t.cpu().tolist()                              # (also drains the process group's stream: collectives of the group's and of
        if on_device:                                       # libsimq's communicator must never be in flight together -- different
            torch.cuda.synchronize(self.device)             # ranks may schedule them in different orders and deadlock)
        if got[0] != 1:
            raise SimqError('Comm: rank 0 could not draw an RCCL identifier%s' % ((': ' + why) if why else ''))
        ident = (ctypes.c_ubyte * COMM_ID_BYTES)(*got[1:])
        h = ctypes.c_void_p()
        try:
            with torch.cuda.device(self.device):
                lib.call('simq_comm_init', ident, self.world, self.rank, ctypes.byref(h))
        except Exception as ex:                              # noqa: BLE001
            status, why, h = 0, str(ex), ctypes.c_void_p()
        if on_device:
            torch.cuda.synchronize(self.device)
        ok = torch.tensor([status], dtype=torch.int32, device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            if h:
                lib.call('simq_comm_destroy', h)
            raise SimqError('Comm: simq_comm_init failed on %s%s' % ('this rank' if status == 0 else 'another rank', (': ' + why) if why else ''))
        self.handle = h

    def world_size(self):
        """Ranks RCCL itself reports for the communicator (simq_comm_world_size)."""
        return int(self._lib.c.simq_comm_world_size(self.handle))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce(self, tensor):
        from ._lib import COMM_F32, COMM_F64, SimqError
        if tensor.dtype not in (torch.float32, torch.float64) or not tensor.is_contiguous() or tensor.device != self.device:
            raise SimqError('Comm.all_reduce: contiguous fp32 / fp64 tensor on %s expected' % self.device)
        self._lib.call('simq_comm_allreduce', self.handle, ctypes.c_void_p(tensor.data_ptr()), tensor.numel(),
                       COMM_F32 if tensor.dtype == torch.float32 else COMM_F64, self._stream())
        return tensor

    def broadcast(self, tensor, src=0):
        from ._lib import SimqError
        if not tensor.is_contiguous() or tensor.device != self.device:
            raise SimqError('Comm.broadcast: contiguous tensor on %s expected (a permuted view would send the wrong bytes)' % self.device)
        if not (0 <= int(src) < self.world):
            raise SimqError('Comm.broadcast: src %r outside [0, %d)' % (src, self.world))
        self._lib.call('simq_comm_broadcast', self.handle, ctypes.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size(),
                       int(src), self._stream())
        return tensor

    def wait(self):
        self._lib.call('simq_comm_wait', self.handle, self._stream())

    def adopt_stream(self, stream):
        """Run the collectives on `stream` (a torch stream the caller has tested not to share the launch stream's hardware queue:
        simq.learner.LearnerStreams.third) instead of the communicator's own; simq_comm_adopt_stream."""
        key = None if stream is None else stream.cuda_stream
        if getattr(self, '_adopted', 'unset') != key:
            self._lib.call('simq_comm_adopt_stream', self.handle, None if stream is None else ctypes.c_void_p(stream.cuda_stream))
            self._adopted, self._adopted_ref = key, stream

    def time_waits(self, on=True):
        """Bracket every simq_comm_wait with timing events (simq_comm_time_waits): exposed communication time of the data-parallel step."""
        self._lib.call('simq_comm_time_waits', self.handle, 1 if on else 0)

    def last_wait_ms(self):
        """Milliseconds the consumer stream stood in the last simq_comm_wait (blocks until that wait has been passed)."""
        ms = ctypes.c_float()
        self._lib.call('simq_comm_last_wait_ms', self.handle, ctypes.byref(ms))
        return float(ms.value)

    def progress(self):
        """{'enqueued', 'completed', 'last'}: collectives this rank has enqueued on the communicator, those the device has finished, and
        what the last one was (simq_comm_progress) -- for a watchdog thread to say WHICH collective a hung rank sits in."""
        out = (ctypes.c_int64 * 4)()
        self._lib.call('simq_comm_progress', self.handle, out)
        kind = {0: 'all-reduce fp32', 1: 'all-reduce fp64', 2: 'broadcast'}.get(int(out[2]), 'none')
        return {'enqueued': int(out[0]), 'completed': int(out[1]), 'last': '%s x %d' % (kind, int(out[3]))}

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.call('simq_comm_destroy', self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
