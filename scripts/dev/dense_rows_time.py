"""Developer: the dense stress table through the UNPLANNED entry points and the plan: sum LL with and without gradient rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
dev = torch.device("cuda", 0)
n = 20_000_000
t = kernels.synth_counts(20211012, 0, n, dev, want=("train",), dense=True)["train"]
prior = kernels.synth_prior(20211012, 0, n, dev)
def timed(fn, reps=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / reps)
    return best
plan = kernels.Plan(t, 5)
a = kernels.dm_prior_planned(plan, prior, -0.3).tolist()
b = kernels.dm_prior(t, prior, -0.3)[0].tolist()
c, g = kernels.dm_prior(t, prior, -0.3, want_grad=True)
pg = kernels.dm_prior_planned(plan, prior, -0.3, want_grad=True)[1]
print("sums planned / unplanned / unplanned+grad:", a, b, c.tolist(), "grad rows max rel diff", float(((g - pg).abs().max() / pg.abs().max())))
print("planned            %.3f ms" % timed(lambda: kernels.dm_prior_planned(plan, prior, -0.3)))
print("planned + rows     %.3f ms" % timed(lambda: kernels.dm_prior_planned(plan, prior, -0.3, want_grad=True)))
print("unplanned (sorted) %.3f ms" % timed(lambda: kernels.dm_prior(t, prior, -0.3)))
print("unplanned + rows   %.3f ms" % timed(lambda: kernels.dm_prior(t, prior, -0.3, want_grad=True)))
pa = kernels.Plan(t, 5, rows_if_dense=True)
print("auto plan: rowwise", pa.rowwise, "bytes per context", pa.nbytes / n, "(sorted form: %.1f)" % (plan.nbytes / n))
print("sums dense form:", kernels.dm_prior_planned(pa, prior, -0.3).tolist())
gd = kernels.dm_prior_planned(pa, prior, -0.3, want_grad=True)[1]
print("grad rows dense form vs sorted form, max rel diff", float((gd - pg).abs().max() / pg.abs().max()))
print("dense form         %.3f ms" % timed(lambda: kernels.dm_prior_planned(pa, prior, -0.3)))
print("dense form + rows  %.3f ms" % timed(lambda: kernels.dm_prior_planned(pa, prior, -0.3, want_grad=True)))
print("dense form, AR     %.3f ms" % timed(lambda: kernels.dm_prior_planned(pa, prior, -0.3, train_ar=True)))
import numpy as np
rf = kernels.synth_counts(20211012, 0, n, dev, want=("ref",), dense=True)["ref"]
argsr = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
pr = kernels.Plan(t, 4, ref=rf)
print("mode R planned (reference-aware) %.3f ms, %.1f B per context" % (timed(lambda: kernels.dm_ref_planned(pr, rf, *argsr)), pr.nbytes / n))
ps = kernels.Plan(t, 4)
print("mode R planned (streaming)       %.3f ms, %.1f B per context" % (timed(lambda: kernels.dm_ref_planned(ps, rf, *argsr)), ps.nbytes / n))
print("mode R unplanned                 %.3f ms" % timed(lambda: kernels.dm_ref(t, rf, *argsr)))
print(kernels.dm_ref_planned(pr, rf, *argsr).tolist(), kernels.dm_ref(t, rf, *argsr).tolist())
