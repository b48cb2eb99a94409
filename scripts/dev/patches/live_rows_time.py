"""Developer: the gradient-row kernel on the full 1e8 table, all rows written against the rows of the contexts with counts only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
dev=torch.device("cuda",0)
n=100_000_000
t=kernels.synth_counts(20211012,0,n,dev,want=("train",))["train"]
prior=kernels.synth_prior(20211012,0,n,dev)
plan=kernels.Plan(t,5)
rows=plan.live_rows(); print("live", rows.shape[0]/n)
def timed(fn,reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best=1e9
    for _ in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/reps)
    return best
o,g=kernels.dm_prior_planned(plan,prior,-0.3,want_grad=True,normalized=True)
ol,gl=kernels.dm_prior_planned(plan,prior,-0.3,want_grad=True,normalized=True,live_only=True)
print("equal", torch.equal(gl,g.index_select(0,rows)), o.tolist(), ol.tolist())
del g,gl
print("full ms", timed(lambda: kernels.dm_prior_planned(plan,prior,-0.3,want_grad=True,normalized=True)))
print("live ms", timed(lambda: kernels.dm_prior_planned(plan,prior,-0.3,want_grad=True,normalized=True,live_only=True)))
