"""Held-out evaluation kernels on the synthetic k=13 table: planned (sorted test-column plan) vs unplanned, kernel-only times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from bear_amd import kernels

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda")
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "test"))
f = kernels.synth_prior(20211012, 0, n, dev)
torch.cuda.synchronize()
t0 = time.time()
plan = kernels.EvalPlan(t["test"], t["train"])
torch.cuda.synchronize()
print(f"plan build {time.time() - t0:.4f} s, {plan.nbytes / n:.2f} B/context")
# as evaluation() holds a batch: the contexts with held-out counts only, their table rows as row_ids
keep = (t["test"] != 0).any(dim=1).nonzero().squeeze(1)
f_k = f.index_select(0, keep).contiguous()
plan_k, ids_k = kernels.EvalPlan(t["test"].index_select(0, keep).contiguous(), t["train"].index_select(0, keep).contiguous()), keep.to(torch.int32)


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, hs, van in (("1 h + AR + 3 van_reg", [1.0], [0.1, 1.0, 10.0]), ("1 h + AR", [1.0], None), ("AR + 3 van", None, [0.1, 1.0, 10.0]),
                      ("h_scan 16", list(range(1, 17)), [1.0])):
    ms_p = timed(lambda: kernels.evaluate_planned(plan, f, hs, van))
    ms_k = timed(lambda: kernels.evaluate_planned(plan_k, f_k, hs, van, row_ids=ids_k))
    ms_u = timed(lambda: kernels.evaluate(t["test"], f, hs, van, t["train"]), 2)
    print(f"{name:24s} compacted {ms_k:8.3f} ms ({80 * n / ms_k / 1e6 / 8000:.3f}) | planned {ms_p:8.3f} ms = {n / ms_p / 1e6:7.2f} Gctx/s ({80 * n / ms_p / 1e6 / 8000:.3f} of HBM peak) | unplanned {ms_u:8.3f} ms = {n / ms_u / 1e6:6.2f} Gctx/s")
