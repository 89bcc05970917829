"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards with no data-path exchange: reads are independent, rank r owns the contiguous
shard `shard_range(n, r, world)` of the input (or its own batches), and the only collective is
the sum of the additive counter buffers at the end of the run -- what Stats::merge /
FilterResult::merge do serially in the reference (src/stats.cpp:1013-1082,
src/filterresult.cpp:28-61).  Ranks first agree on the per-cycle capacity C (a 1-element MAX
all-reduce) so that the cycle-major buffers line up."""
import numpy as np

from . import abi


def shard_range(n_items, rank, world):
    """contiguous, balanced [begin, end) of rank's shard"""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def agree_capacity(local_max_cycles, device=None, group=None):
    import torch
    import torch.distributed as dist

    t = torch.tensor([int(local_max_cycles)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def allreduce_counters(counters_t, group=None):
    """in-place SUM of the int64 counter buffer over all ranks (RCCL over xGMI on GPUs)"""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counters_t, op=dist.ReduceOp.SUM, group=group)
    return counters_t


def merge_host_counters(buf, c_local, nad, group=None):
    """numpy variant used by hosts that hold their counters in host memory: regrid to the agreed
    capacity, all-reduce, return (merged buffer, C)"""
    import torch

    c = agree_capacity(c_local, group=group)
    if c != c_local:
        buf = abi.regrid_counters(buf, c_local, c, nad)
    t = torch.from_numpy(np.ascontiguousarray(buf))
    allreduce_counters(t, group=group)
    return t.numpy(), c
