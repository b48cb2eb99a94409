"""Dev helper: time bear_eval_f64 / bear_bmm_f64 on the synthetic table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(os.environ.get("N", "1e8")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "test"))
f = kernels.synth_prior(20211012, 0, N, dev)
van = [0.1, 1.0, 10.0]
res = []
for name, fn in [("eval H=1 V=3", lambda: kernels.evaluate(t["test"], f, [1.0], van, t["train"])),
                 ("eval H=1 V=3 no-train", lambda: kernels.evaluate(t["test"], f, [1.0], van, None)),
                 ("h_scan H=16 V=1", lambda: kernels.evaluate(t["test"], f, np.geomspace(0.01, 100, 16), [1.0], t["train"], with_ar=False)),
                 ("bmm V=3", lambda: kernels.bmm(t["train"], van))]:
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): out = fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    res.append(f"{name}: {best:.3f} ms ({N / best / 1e6:.1f} Gctx/s)")
print(" | ".join(res))
