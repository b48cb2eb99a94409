"""Dev helper: the two single-buffered gradient-row kernels (dm_prior_plan_grad_kernel: rows not asserted normalised;
dm_refmix_plan_grad_kernel: bear_ref's step with the mixing inside) at 1e8 contexts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels
n = int(float(os.environ.get("N", "1e8")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "ref"))
prior = kernels.synth_prior(20211012, 0, n, dev)
plan = kernels.Plan(t["train"], 5)


def timed(fn, reps=10):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


print("general gradient rows  %.3f ms" % timed(lambda: kernels.dm_prior_planned(plan, prior, 0.0, want_grad=True)))
print("normalised (in place)  %.3f ms" % timed(lambda: kernels.dm_prior_planned(plan, prior, 0.0, want_grad=True, normalized=True)))
ref_in = t["ref"].to(torch.float64) + 1e-7
ref_in[:, -1] = 0
z = lambda v: torch.tensor([v] if np.ndim(v) == 0 else v, dtype=torch.float64, device=dev)
h, tau, nw = z(0.0), torch.tensor(float(np.log(1 / 30)), dtype=torch.float64, device=dev), torch.tensor(float(-np.log(100)), dtype=torch.float64, device=dev)
print("refmix fused step      %.3f ms" % timed(lambda: kernels.dm_refmix_planned_dev(plan, prior, ref_in, h, tau, nw)))
