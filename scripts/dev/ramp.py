"""Dev helper: per-launch duration of the headline kernel from a cold start (clock / power-state ramp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
N = 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
plan = kernels.Plan(t["train"], 5)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(301)]
evs[0].record()
for k in range(300):
    kernels.dm_prior_planned(plan, f, 0.0)
    evs[k + 1].record()
torch.cuda.synchronize()
d = [evs[k].elapsed_time(evs[k + 1]) for k in range(300)]
for a in range(0, 300, 20):
    print(a, " ".join("%.3f" % x for x in d[a:a + 20]))
