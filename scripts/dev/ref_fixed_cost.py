"""Developer: the mode-R step (bear_dm_ref_plan_f64 on a reference-aware plan: dm_ref_items_kernel) against the table size -- what a launch costs whatever its size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import ctypes
from bear_amd import kernels, _lib
dev = torch.device("cuda", 0)
args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
for n in [int(float(a)) for a in sys.argv[1:]] or [2000, 100_000, 1_000_000, 10_000_000]:
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "ref"))
    plan = kernels.Plan(t["train"], 4, ref=t["ref"])
    fn = lambda: kernels.dm_ref_planned(plan, t["ref"], *args)
    for _ in range(50): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{n:>9d} contexts: {best * 1e3:8.1f} us per launch", flush=True)
    L = _lib.lib()
    if hasattr(L, "bear_dbg_ref_pe_stamps"):        # -DBEAR_DEV_BUILD -DLIN_STAMPS
        buf = (ctypes.c_ulonglong * 8)()
        L.bear_dbg_ref_pe_stamps(buf)
        names = ["params+log table", "item units", "hist0+lists", "big totals", "stop+small hists+block sums", "last arrival", "partials sum", "update"]
        print("      " + "  ".join(f"{nm} {buf[k]}" for k, nm in enumerate(names)), flush=True)
