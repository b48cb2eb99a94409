"""Row sharding, process-group set-up and the one collective of a training step.

K-mer contexts are independent, so a table shards by contiguous row ranges, one process per GPU; each
step every rank reduces its shard to the packed fp64 vector ``[sum LL, grads...]`` and a single
``all_reduce(sum)`` combines them (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
This replaces ``tf.distribute.MirroredStrategy`` (bear_net.py:246): ``strategy.reduce`` (bear_net.py:290),
the cross-replica gradient sum inside ``optimizer.apply_gradients`` (bear_net.py:278-282), the mirrored
variables (``broadcast_params``) and ``experimental_distribute_dataset`` (bear_net.py:273; ``shard_rows``).

Launch: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
bear_amd/models/train_bear_ref.py config.cfg`` -- the drivers call ``init_from_env()`` before any GPU work.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env():
    """One process per GPU: binds this process to its device and joins the process group when the launcher
    (``torch.distributed.run``) exported WORLD_SIZE > 1.  Must run before anything touches the GPU.

    Environment: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT from the launcher;
    ``BEAR_AMD_DIST_BACKEND`` (default ``nccl`` = RCCL; ``gloo`` for tests that put several ranks on one GPU);
    ``BEAR_AMD_DEVICE`` overrides the device index (default LOCAL_RANK).
    Returns ``(rank, world_size)``."""
    if dist.is_available() and dist.is_initialized():
        return world()
    # dmabuf IPC: RCCL between processes (and CUDA-tensor sharing) needs it on this host driver -- without it the first
    # collective of an externally launched job fails in hipIpcGetMemHandle.  The HSA runtime reads the variable when the GPU is
    # first touched, which is why this function must run before any GPU call.  A value the launcher exported wins.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = int(os.environ.get("BEAR_AMD_DEVICE", local_rank))
    if torch.cuda.device_count() > dev_index:   # device_count() does not initialise the GPU on this image
        torch.cuda.set_device(dev_index)
    if world_size <= 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("BEAR_AMD_DIST_BACKEND", "nccl")
    rank = int(os.environ["RANK"])
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", dev_index))
    else:
        if backend == "gloo" and os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
            # one node: gloo otherwise resolves the host name to pick its interface (the name may not resolve: a silent hang)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group(backend, rank=rank, world_size=world_size)
    return world()


def shutdown(barrier=True):
    """Leaves the process group (no-op without one).  ``barrier=False`` on error paths: the other ranks may never arrive."""
    if dist.is_available() and dist.is_initialized():
        if barrier:
            dist.barrier()
        dist.destroy_process_group()


def shard_rows(n_rows, rank=None, world_size=None):
    """Contiguous, balanced row range [lo, hi) of `rank`; the first n_rows % world ranks get one more row."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def collective_active():
    """True when a step's all-reduce actually runs: several ranks -- or a group of ONE with BEAR_AMD_COLLECTIVE_ALWAYS=1, which
    sends the packed vector through the backend anyway (the only way to exercise RCCL itself on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or bool(os.environ.get("BEAR_AMD_COLLECTIVE_ALWAYS"))


def collective_capturable():
    """Can the step's all-reduce be captured into a HIP graph?  Yes without one, yes on RCCL (stream-ordered), no on gloo (host)."""
    return (not collective_active()) or dist.get_backend() == "nccl"


def _via_host(t):
    """gloo moves host memory: CUDA tensors are staged through the host for it (tests only; RCCL reduces in place)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def allreduce_sum_(packed):
    """In-place sum of the packed per-shard partials over all ranks (no-op for a single process).  On RCCL this is
    enqueued on the current stream: no host synchronisation."""
    if collective_active():
        if _via_host(packed):
            h = packed.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            packed.copy_(h)
        else:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    return packed


def allreduce_max_(t):
    if world()[1] > 1:
        if _via_host(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def agree_max(value, device=None):
    """The largest ``value`` (an int) any rank holds: every rank calls it and every rank gets the same answer, so a decision
    that a rank would otherwise take from its OWN state (free HBM, an environment switch) and that changes which collectives
    it issues is taken by all of them alike.  One process: the value itself."""
    if world()[1] <= 1:
        return int(value)
    on = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def broadcast_params(params, src=0):
    """Mirrored variables: every rank starts from rank `src`'s parameter values (MirroredStrategy creates variables once
    and mirrors them, bear_net.py:248-256; without this each rank would keep the draws of its own RNG)."""
    if world()[1] <= 1:
        return params
    with torch.no_grad():
        for p in params:
            if _via_host(p):
                h = p.detach().cpu()
                dist.broadcast(h, src=src)
                p.copy_(h)
            else:
                dist.broadcast(p.data, src=src)
    return params


def barrier():
    if world()[1] > 1:
        dist.barrier()


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
