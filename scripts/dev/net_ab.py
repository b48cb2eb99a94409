"""Dev helper: the headline mode-N planned kernel, timed with the library named by BEAR_AMD_LIB (A/B runs on one box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
N = 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
plan = kernels.Plan(t["train"], 5)
out = kernels.dm_prior_planned(plan, f, 0.0).cpu().numpy()
for name, kw in (("general", {}), ("normalized", {"normalized": True})):
    fn = lambda: kernels.dm_prior_planned(plan, f, 0.0, **kw)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20)
    print(os.environ.get("BEAR_AMD_LIB", "default").split("/")[-1], name, " ".join("%.4f" % b for b in best), "ms", out)
