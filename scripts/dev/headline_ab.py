"""Dev helper: the headline kernel (mode N, 1e8 contexts) and the paired linear step, event-timed after a warm-up: A/B of library builds (BEAR_AMD_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
dev = torch.device("cuda", 0)
n = 100_000_000
t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
prior = kernels.synth_prior(20211012, 0, n, dev)
plan = kernels.Plan(t, 5)
out = torch.zeros(2, dtype=torch.float64, device=dev)
fn = lambda: kernels.dm_prior_planned(plan, prior, 0.0, out=out)
for _ in range(200): fn()
torch.cuda.synchronize()
best = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 50)
print(os.environ.get("BEAR_AMD_LIB", "default").split("/")[-1], "headline: min %.4f median %.4f ms" % (min(best), sorted(best)[len(best) // 2]), out.tolist())
