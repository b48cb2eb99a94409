"""Dev helper: the planned mode-N kernel that also writes the gradient rows (any torch AR function trains through it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
N = int(float(os.environ.get("N", "1e8")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
plan = kernels.Plan(t["train"], 5)


def timed(fn, reps=10):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.empty_like(f)
for norm, ar in ((False, False), (True, False), (True, True)):
    ms = timed(lambda: kernels.dm_prior_planned(plan, f, 0.0, want_grad=True, normalized=norm, train_ar=ar))
    print(f"gradient rows  normalized={norm} ar={ar}: {ms:.3f} ms  ({N * 84 / ms / 1e9:.2f} TB/s on 44 + 40 B per context)")
ms = timed(lambda: kernels.dm_prior_planned(plan, f, 0.0))
print(f"without gradient rows: {ms:.3f} ms")
