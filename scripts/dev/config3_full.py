"""Dev helper: BASELINE configs[3] at its FULL size on one GPU (1e9 contexts, mode R): the whole table == the sum of the eight
rank shards of the 8-GPU job (each with its own plan), and the per-step kernel time at that size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels
dev = torch.device("cuda", 0)
N, W = 1_000_000_000, 8
args = (0.2, float(np.log(1 / 30)), float(-np.log(100)))
t0 = time.time()
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
torch.cuda.synchronize(); t1 = time.time()
plan = kernels.Plan(t["train"], 4, ref=t["ref"])
torch.cuda.synchronize(); t2 = time.time()
whole = kernels.dm_ref_planned(plan, t["ref"], *args).cpu().numpy()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(20): kernels.dm_ref_planned(plan, t["ref"], *args)
e0.record()
for _ in range(50): kernels.dm_ref_planned(plan, t["ref"], *args)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("1e9 contexts: synth %.2f s, plan %.2f s (%.2f B/context), step %.3f ms = %.0f Gctx/s" % (t1 - t0, t2 - t1, plan.nbytes / N, ms, N / ms / 1e6), flush=True)
del plan
parts = np.zeros(4)
n = N // W
for r in range(W):
    tr, rf = t["train"][r * n:(r + 1) * n], t["ref"][r * n:(r + 1) * n]
    parts += kernels.dm_ref_planned(kernels.Plan(tr, 4, ref=rf), rf, *args).cpu().numpy()
print("whole", whole, "sum of 8 shards", parts, "rel diff", np.abs(whole - parts) / np.abs(whole))
