"""Row sharding and the one collective of a training step.

K-mer contexts are independent, so a table shards by contiguous row ranges, one process per GPU; each
step every rank reduces its shard to the packed fp64 vector ``[sum LL, grads...]`` and a single
``all_reduce(sum)`` combines them (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
This replaces ``strategy.reduce`` (bear_net.py:290) and the cross-replica gradient sum inside
``optimizer.apply_gradients`` (bear_net.py:278-282).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rows(n_rows, rank=None, world_size=None):
    """Contiguous, balanced row range [lo, hi) of `rank`; the first n_rows % world ranks get one more row."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_sum_(packed):
    """In-place sum of the packed per-shard partials over all ranks (no-op for a single process)."""
    if world()[1] > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    return packed


def pack(tensors):
    """Flattens a list of tensors into one fp64 vector and returns (flat, unpack)."""
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])
    shapes = [t.shape for t in tensors]

    def unpack(v):
        out, k = [], 0
        for s in shapes:
            m = int(torch.Size(s).numel())
            out.append(v[k:k + m].reshape(s))
            k += m
        return out
    return flat, unpack
