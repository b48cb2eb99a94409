"""Dev helper: time the two hot-path kernels on a synthetic table (not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dense = len(sys.argv) > 2 and sys.argv[2] == "dense"
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, dense=dense, want=("train", "ref"))
f = kernels.synth_prior(20211012, 0, N, dev)
torch.cuda.synchronize()
print("n mean", t["train"].sum(1).double().mean().item(), "P(n=0)", (t["train"].sum(1) == 0).double().mean().item(),
      "max", t["train"].max().item(), "ref mean", t["ref"].sum(1).double().mean().item())
args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
for name, fn, bpr in [("ref", lambda: kernels.dm_ref(t["train"], t["ref"], *args), 40),
                      ("prior", lambda: kernels.dm_prior(t["train"], f, 0.0)[0], 60),
                      ("prior+grad", lambda: kernels.dm_prior(t["train"], f, 0.0, want_grad=True)[0], 60),
                      ("ref_ar", lambda: kernels.dm_ref(t["train"], t["ref"], *args, train_ar=True), 40)]:
    out = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:12s} {ms:9.3f} ms  {N/ms/1e6:9.3f} Gctx/s  {N*bpr/ms/1e6:9.1f} GB/s  out={out.cpu().numpy()}")
