"""Multi-GPU MSM plumbing: one process per GPU, the pairs sharded by contiguous range (the
partition best_multiexp itself uses across rayon threads, arithmetic.rs:137-152), one RCCL
all-gather of the 96-byte Jacobian partials, then the left fold of arithmetic.rs:153 on every
rank.  RCCL has no elliptic-curve reduction operator, so "reduce" = gather bytes + fold.
torch.distributed is the transport only (backend "nccl" is RCCL on ROCm; "gloo" in CPU tests)."""
import numpy as np

_bufs = {}


def shard_range(n, rank, world):
    """contiguous shard [lo, hi) of n pairs for `rank`; the last rank takes the remainder"""
    per = n // world
    lo = rank * per
    hi = n if rank == world - 1 else lo + per
    return lo, hi


def allgather_fold(partial_xyz, h2, device=None, group=None):
    """partial_xyz: (12,) uint64 Jacobian partial of this rank -> (12,) uint64 fold over all ranks,
    identical on every rank.  One collective (all_gather_into_tensor of 12 int64 per rank); with a device
    (RCCL) the 96 bytes go up and the gathered bytes come down through pinned host tensors (no pageable
    staging copies); the staging tensors are reused."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    key = (str(device), world, id(group))
    if key not in _bufs:
        dev = device if device is not None else "cpu"
        pin = device is not None
        _bufs[key] = (torch.empty(12, dtype=torch.int64, device=dev), torch.empty(12 * world, dtype=torch.int64, device=dev),
                      torch.empty(12, dtype=torch.int64, pin_memory=pin), torch.empty(12 * world, dtype=torch.int64, pin_memory=pin))
    mine, gathered, h_mine, h_all = _bufs[key]
    h_mine.numpy()[:] = np.ascontiguousarray(partial_xyz, dtype=np.uint64).view(np.int64)
    if device is None:
        dist.all_gather_into_tensor(h_all, h_mine, group=group)
    else:
        mine.copy_(h_mine, non_blocking=True)
        dist.all_gather_into_tensor(gathered, mine, group=group)
        h_all.copy_(gathered, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    parts = h_all.numpy().view(np.uint64).reshape(world, 12)
    return h2.g1_fold(parts)


def allgather_start(partial_xyz, device=None, group=None, slot=0):
    """Start the all-gather of this rank's 96-byte partial and return a handle for allgather_finish.  With a device
    (RCCL) the collective is asynchronous (`async_op=True`): the caller may run the next shard MSM while the bytes
    travel; two staging slots alternate so that a pending gather's buffers are not overwritten."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    key = (str(device), world, id(group), "async", slot & 1)
    if key not in _bufs:
        dev = device if device is not None else "cpu"
        pin = device is not None
        _bufs[key] = (torch.empty(12, dtype=torch.int64, device=dev), torch.empty(12 * world, dtype=torch.int64, device=dev),
                      torch.empty(12, dtype=torch.int64, pin_memory=pin), torch.empty(12 * world, dtype=torch.int64, pin_memory=pin))
    mine, gathered, h_mine, h_all = _bufs[key]
    h_mine.numpy()[:] = np.ascontiguousarray(partial_xyz, dtype=np.uint64).view(np.int64)
    if device is None:
        work = dist.all_gather_into_tensor(h_all, h_mine, group=group, async_op=True)
        return (work, None, h_all, world, None)
    mine.copy_(h_mine, non_blocking=True)
    work = dist.all_gather_into_tensor(gathered, mine, group=group, async_op=True)
    return (work, gathered, h_all, world, device)


def allgather_finish(handle, h2):
    """wait for the gather started by allgather_start and fold the partials (arithmetic.rs:153): (12,) uint64"""
    import torch
    work, gathered, h_all, world, device = handle
    work.wait()
    if device is not None:
        h_all.copy_(gathered, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    return h2.g1_fold(h_all.numpy().view(np.uint64).reshape(world, 12))
