"""Developer: mode-R planned kernel time only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
f = kernels.synth_prior(20211012, 0, N, dev)
args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
plan_r, plan_n = kernels.Plan(t["train"], 4), kernels.Plan(t["train"], 5)
import time
t0 = time.time(); plan_rr = kernels.Plan(t["train"], 4, ref=t["ref"]); torch.cuda.synchronize(); print("ref-aware plan build %.3f s, %.2f B/context" % (time.time() - t0, plan_rr.nbytes / N))
for name, fn in (("ref_plan", lambda: kernels.dm_ref_planned(plan_r, t["ref"], *args)), ("ref_aware", lambda: kernels.dm_ref_planned(plan_rr, t["ref"], *args)), ("ref_aware_ar", lambda: kernels.dm_ref_planned(plan_rr, t["ref"], *args, train_ar=True)), ("prior_plan", lambda: kernels.dm_prior_planned(plan_n, f, 0.0)),
                 ("ref_plan_ar", lambda: kernels.dm_ref_planned(plan_r, t["ref"], *args, train_ar=True))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) / 20:8.3f} ms")
