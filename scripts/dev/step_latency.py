"""Dev helper: wall time of ONE optimizer step (reduce -> [all-reduce] -> apply) as `_train.run_device_steps` enqueues it, per
shard size: eager loop vs. HIP-graph replay, with and without the RCCL all-reduce (a process group of ONE rank -- the only RCCL
there is on a one-GPU box; its time is the collective's launch + kernel latency, the floor of what N ranks pay per step).
    python scripts/dev/step_latency.py            # prints a table; numbers for DESIGN.md section 5's latency model"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as tdist

from bear_amd import _train, dist, kernels

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
SEED = 20211012
STEPS = 400


print("%-10s %-8s %12s %12s %12s %12s" % ("contexts", "path", "eager", "eager+rccl", "graph", "graph+rccl"), " (microseconds per step)")
for n in (1365, 100_000, 1_000_000, 12_500_000, 100_000_000):
    t = kernels.synth_counts(SEED, 0, n, dev, want=("train", "ref"))
    plan_r = kernels.Plan(t["train"], 4, ref=t["ref"])
    plan_n = kernels.Plan(t["train"], 5)
    lag = 13
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(1))
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        key = key * 6 + codes[:, l].to(torch.int64)
    packed_k = kernels.linear_index(kernels.pack_kmers(codes[torch.argsort(key)].contiguous()), lag)
    del key, codes
    theta_r = torch.tensor([0.0, np.log(1 / 30), -np.log(100)], dtype=torch.float64, device=dev)
    theta_n = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev),
                         0.05 * torch.randn(lag * 25, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(10))])
    fns = {"ref": (lambda packed: kernels.ref_train_reduce(plan_r, t["ref"], theta_r_live, packed), theta_r),
           "linear": (lambda packed: kernels.net_linear_train_reduce(plan_n, packed_k, lag, theta_n_live, packed), theta_n)}
    for name, (fn, th0) in fns.items():
        row = []
        for mode in ("eager", "graph"):
            for coll in (False, True):
                # run_device_steps updates the theta it is given in place: the reducers read that same tensor
                live = th0.clone()
                if name == "ref":
                    theta_r_live = live
                else:
                    theta_n_live = live
                best = None
                for _ in range(3):
                    live.copy_(th0)
                    if coll:
                        os.environ["BEAR_AMD_COLLECTIVE_ALWAYS"] = "1"
                    else:
                        os.environ.pop("BEAR_AMD_COLLECTIVE_ALWAYS", None)
                    if mode == "eager":
                        os.environ["BEAR_AMD_NO_GRAPH"] = "1"
                    else:
                        os.environ.pop("BEAR_AMD_NO_GRAPH", None)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _train.run_device_steps([fn], [-1.0 / n], live, STEPS, 1e-3, "Adam", False, 1, dev)
                    dt = (time.perf_counter() - t0) / STEPS
                    best = dt if best is None else min(best, dt)
                row.append(best * 1e6)
        print("%-10d %-8s %12.1f %12.1f %12.1f %12.1f" % (n, name, row[0], row[1], row[2], row[3]), flush=True)
    del t, plan_r, plan_n, packed_k
tdist.destroy_process_group()
