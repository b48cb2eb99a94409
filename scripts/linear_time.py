"""Dev helper: the linear-head training step, fused kernel vs. torch ar_func + planned gradient-row kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels, ar_funcs
N = int(float(os.environ.get("N", "1e8")))
LAG = int(os.environ.get("LAG", "13"))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev)
packed = kernels.pack_kmers(codes)
plan = kernels.Plan(t["train"], 5)
torch.manual_seed(0)
f, (mat,) = ar_funcs.make_ar_func_linear(LAG, 4, device=dev)
res = []
def fused(ar=False):
    return kernels.dm_linear(plan, packed, mat.detach(), 0.0, train_ar=ar)
def unfused():
    mat.grad = None
    prior = f(codes)
    out, g = kernels.dm_prior_planned(plan, prior.detach(), 0.0, want_grad=True)
    prior.backward(g)
    return out, mat.grad
cases = [("fused", fused), ("fused_ar", lambda: fused(True))]
if N <= 20_000_000:
    cases.append(("torch+planned_grad", unfused))
for name, fn in cases:
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): out = fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    res.append(f"{name}: {best:.3f} ms ({N / best / 1e6:.2f} Gctx/s)")
print(f"N={N} lag={LAG} | " + " | ".join(res))
if N <= 20_000_000:
    a, b = fused(), unfused()
    print("max |d mat| diff", (a[1] - b[1]).abs().max().item(), "of", b[1].abs().max().item(), "elbo", a[0][0].item(), b[0][0].item())
