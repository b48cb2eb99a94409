"""Profiling driver: builds the synthetic table and launches each hot-path kernel a few times.
Usage (on the GPU box):  rocprofv3 --kernel-trace --stats ... -- python3 scripts/prof_driver.py 1e8 [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
f = kernels.synth_prior(20211012, 0, N, dev) if mode in ("both", "prior") else None
args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
torch.cuda.synchronize()
planned = os.environ.get("BEAR_PROF_UNPLANNED") is None
if planned:
    plan_n = kernels.Plan(t["train"], 5) if mode in ("both", "prior") else None
    plan_r = kernels.Plan(t["train"], 4) if mode in ("both", "ref") else None
for _ in range(reps):
    if mode in ("both", "prior"):
        kernels.dm_prior_planned(plan_n, f, 0.0) if planned else kernels.dm_prior(t["train"], f, 0.0)
    if mode in ("both", "ref"):
        kernels.dm_ref_planned(plan_r, t["ref"], *args) if planned else kernels.dm_ref(t["train"], t["ref"], *args)
torch.cuda.synchronize()
print("done")
