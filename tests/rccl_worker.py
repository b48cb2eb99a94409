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
    # the product optimizer loop with the RCCL all-reduce INSIDE the captured HIP graph (BEAR_AMD_COLLECTIVE_ALWAYS=1 sends the
    # packed vector through the backend although the group has one rank): captured + replayed == enqueued eagerly
    from bear_amd import _train, ar_funcs, bear_net, bear_ref, dataloader
    ysd1 = os.path.join(os.environ["BEAR_ROOT"], "tests", "golden", "ysd1_lag_5_file_0_preshuf.tsv")
    data = dataloader.dataloader(ysd1, "dna", 500, 3)
    os.environ["BEAR_AMD_COLLECTIVE_ALWAYS"] = "1"
    runs = {}
    for name, fn, args, kw in (
            ("ref", bear_ref.train, (data.repeat(6), 1365, 6, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False), {}),
            ("net_acc2", bear_net.train, (data.repeat(6), 1365, 6, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False),
             {"acc_steps": 2}),
            # an AR function of torch ops (a cnn shape the fused kernels do not take): the generic loop, its all-reduce in the graph too
            ("net_generic", bear_net.train, (data.repeat(6), 1365, 6, 0, "dna", 5, ar_funcs.make_ar_func_cnn,
                                             {"num_filters": 20, "filter_width": 3, "kmer_layer1_width": 16}, 0.01, "Adam", False), {})):
        for mode in ("graph", "eager"):
            if mode == "eager":
                os.environ["BEAR_AMD_NO_GRAPH"] = "1"
            torch.manual_seed(5)
            ls = []
            p, _, _ = fn(*args, loss_save=ls, **kw)
            os.environ.pop("BEAR_AMD_NO_GRAPH", None)
            runs[name + "_" + mode] = {"loss": ls, "params": torch.cat([x.detach().reshape(-1) for x in p]).cpu().tolist(),
                                       "how": dict(_train.LAST_RUN)}
    res["runs"] = runs
    dist.destroy_process_group()
    json.dump(res, open(os.environ["BEAR_OUT"], "w"))


if __name__ == "__main__":
    main()
