"""Dev helper: time the hot-path kernels on a synthetic table (not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from bear_amd import kernels
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dense = len(sys.argv) > 2 and sys.argv[2] == "dense"
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, dense=dense, want=("train", "ref"))
f = kernels.synth_prior(20211012, 0, N, dev)
torch.cuda.synchronize()
args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
import c_oracle as co
M = min(N, 2_000_000)
tr = t["train"][:M].cpu().numpy().view(np.uint32); rf = t["ref"][:M].cpu().numpy().view(np.uint32)
wr = co.dm_ref(tr, rf, *args, nthreads=8); wn, _ = co.dm_prior(tr, f[:M].cpu().numpy(), 0.0, nthreads=8)
gr = kernels.dm_ref(t["train"][:M], t["ref"][:M], *args).cpu().numpy(); gn = kernels.dm_prior(t["train"][:M], f[:M], 0.0)[0].cpu().numpy()
print("rel err ref  ", np.abs(gr - wr) / np.abs(wr))
print("rel err prior", np.abs(gn - wn) / np.abs(wn))
import time as _t
_t0 = _t.time(); plan_r = kernels.Plan(t["train"], 4); plan_n = kernels.Plan(t["train"], 5); torch.cuda.synchronize()
print("plan build (both) %.3f s; bytes/row: ref %.2f net %.2f" % (_t.time() - _t0, plan_r.nbytes / N, plan_n.nbytes / N))
gpr = kernels.dm_ref_planned(plan_r, t["ref"], *args).cpu().numpy(); gpn = kernels.dm_prior_planned(plan_n, f, 0.0).cpu().numpy()
wr_full = kernels.dm_ref(t["train"], t["ref"], *args).cpu().numpy(); wn_full = kernels.dm_prior(t["train"], f, 0.0)[0].cpu().numpy()
print("planned vs unplanned rel diff", np.abs(gpr - wr_full) / np.abs(wr_full), np.abs(gpn - wn_full) / np.abs(wn_full))
for name, fn, bpr in [("ref_plan", lambda: kernels.dm_ref_planned(plan_r, t["ref"], *args), 40),
                      ("prior_plan", lambda: kernels.dm_prior_planned(plan_n, f, 0.0), 60),
                      ("ref", lambda: kernels.dm_ref(t["train"], t["ref"], *args), 40),
                      ("prior", lambda: kernels.dm_prior(t["train"], f, 0.0)[0], 60),
                      ("prior+grad", lambda: kernels.dm_prior(t["train"], f, 0.0, want_grad=True)[0], 60),
                      ("ref_ar", lambda: kernels.dm_ref(t["train"], t["ref"], *args, train_ar=True), 40)]:
    out = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:12s} {ms:9.3f} ms  {N/ms/1e6:9.3f} Gctx/s  {N*bpr/ms/1e6:9.1f} GB/s  out={out.cpu().numpy()}")
