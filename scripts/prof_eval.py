"""Profiling driver for the evaluation kernels: rocprofv3 ... -- python3 scripts/prof_eval.py 2e7 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import kernels
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "test"))
f = kernels.synth_prior(20211012, 0, n, dev)
plan = kernels.EvalPlan(t["test"], t["train"])
for _ in range(reps):
    kernels.evaluate_planned(plan, f, [1.0], [0.1, 1.0, 10.0])
torch.cuda.synchronize()
print("done")
