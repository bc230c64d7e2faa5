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

This module only uses torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests)
and never touches the HIP library, so the sharding / reduction logic is testable without a GPU.
"""
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
