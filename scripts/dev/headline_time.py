"""Dev helper: the headline kernel (dm_prior_plan_kernel, three instantiations) at steady clocks: median of event-timed groups,
with the sums printed (A/B builds through BEAR_AMD_LIB must agree to the last digits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(os.environ.get("N", "1e8")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
plan = kernels.Plan(t["train"], 5)


def timed(fn, groups=9, per=16):
    for _ in range(150): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    for a, b in evs:
        a.record()
        for _ in range(per): fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) / per for a, b in evs)
    return ms[len(ms) // 2], ms[0]


for name, kw in (("general", {}), ("normalized", {"normalized": True}), ("multinomial", {"train_ar": True})):
    out = kernels.dm_prior_planned(plan, f, -0.3, **kw).cpu().numpy()
    med, best = timed(lambda: kernels.dm_prior_planned(plan, f, -0.3, **kw))
    print(f"{name:12s} median {med:.4f} ms  best {best:.4f} ms  ({N * 44 / med / 1e9:.2f} TB/s on the 44 B moved)  sums {out[0]:.15e} {out[1]:.15e}", flush=True)
med, _ = timed(lambda: kernels.stream_read(f))
print(f"stream read of the prior rows: {med:.4f} ms ({N * 40 / med / 1e9:.2f} TB/s)")
