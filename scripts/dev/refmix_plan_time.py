"""Dev helper: bear_ref's step for a net function with parameters on 1e8 contexts -- the mixing inside the DM kernel
(bear_dm_refmix_plan_grad_f64) against the three launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(os.environ.get("N", "1e8")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
g = kernels.synth_prior(1, 0, N, dev)
ref = t["ref"].to(torch.float64) + 1e-7
ref[:, -1] = 0
plan = kernels.Plan(t["train"], 5)
h, tau, nw = [torch.tensor([v], dtype=torch.float64, device=dev) for v in (0.0, np.log(1 / 30), -np.log(100))]


def timed(fn, reps=5):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def unfused():
    f = kernels.ref_mix_forward(g, ref, tau, nw)
    out, q = kernels.dm_prior_planned_dev(plan, f, h, want_grad=True, normalized=True)
    return out, kernels.ref_mix_backward(g, ref, q, tau, nw)


o1, r1 = kernels.dm_refmix_planned_dev(plan, g, ref, h, tau, nw)
o2, (r2, sc) = unfused()
print("fused  ", o1.cpu().numpy())
print("unfused", o2.cpu().numpy(), sc.cpu().numpy(), " max row diff", float((r1 - r2).abs().max()))
print(f"fused {timed(lambda: kernels.dm_refmix_planned_dev(plan, g, ref, h, tau, nw)):.3f} ms   three launches {timed(unfused):.3f} ms")
