"""Child process of test_dist_gpu.py::test_rccl_group_of_one: joins an RCCL ("nccl") process group of ONE rank on cuda:0 (two ranks
cannot share a card under RCCL) and runs the collectives of a training step behind the planned kernel, in stream order:
all-reduce(sum) of the packed (ELBO, gradient) vector, all-reduce(max) of a timing scalar (bench.py), broadcast of the
parameters (dist.broadcast_params), barrier.  Writes what it saw as JSON."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.environ["BEAR_ROOT"])
from bear_amd import kernels  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n = 200_000
    t = kernels.synth_counts(11, 0, n, dev, want=("train",))
    prior = kernels.synth_prior(11, 0, n, dev)
    plan = kernels.Plan(t["train"], 5)
    out = torch.zeros(2, dtype=torch.float64, device=dev)
    kernels.dm_prior_planned(plan, prior, 0.0, out=out)
    before = out.clone()
    dist.all_reduce(out)                               # enqueued behind the kernel: no host synchronisation in between
    el = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    theta = torch.arange(5, dtype=torch.float64, device=dev)
    dist.broadcast(theta, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "same": bool(torch.equal(before, out)),
           "out": out.cpu().tolist(), "max": float(el.item()), "theta": theta.cpu().tolist()}
    dist.destroy_process_group()
    json.dump(res, open(os.environ["BEAR_OUT"], "w"))


if __name__ == "__main__":
    main()
