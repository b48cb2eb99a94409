"""Dev helper: time planned kernels for one library variant (BEAR_AMD_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels
N = 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
f = kernels.synth_prior(20211012, 0, N, dev)
args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
plan_r = kernels.Plan(t["train"], 4); plan_n = kernels.Plan(t["train"], 5)
res = []
for name, fn in [("ref", lambda: kernels.dm_ref_planned(plan_r, t["ref"], *args)), ("net", lambda: kernels.dm_prior_planned(plan_n, f, 0.0)), ("net_norm", lambda: kernels.dm_prior_planned(plan_n, f, 0.0, normalized=True)), ("net_grad", lambda: kernels.dm_prior_planned(plan_n, f, 0.0, want_grad=True)[0]), ("ref_ar", lambda: kernels.dm_ref_planned(plan_r, t["ref"], *args, train_ar=True)), ("net_ar", lambda: kernels.dm_prior_planned(plan_n, f, 0.0, train_ar=True)), ("net_ar_grad", lambda: kernels.dm_prior_planned(plan_n, f, 0.0, want_grad=True, train_ar=True)[0])]:
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    res.append(f"{name} {best:.3f} ms")
print(os.path.basename(os.environ.get("BEAR_AMD_LIB", "default")), " | ".join(res), out.cpu().numpy()[0])
