"""Dev helper: bear_ref_mix_forward / backward_f64 on 1e8 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels
N = int(float(os.environ.get("N", "1e8")))
dev = torch.device("cuda", 0)
g = kernels.synth_prior(1, 0, N, dev)
q = torch.randn(N, 5, dtype=torch.float64, device=dev)
ref = kernels.synth_counts(20211012, 0, N, dev, want=("ref",))["ref"].to(torch.float64) + 1e-7
ref[:, -1] = 0
t, w = torch.tensor(np.log(1 / 30), dtype=torch.float64, device=dev), torch.tensor(-np.log(100), dtype=torch.float64, device=dev)


def timed(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


f_ms = timed(lambda: kernels.ref_mix_forward(g, ref, t, w))
b_ms = timed(lambda: kernels.ref_mix_backward(g, ref, q, t, w))
print(f"mix forward {f_ms:.3f} ms ({N * 120 / f_ms / 1e9:.2f} TB/s on 120 B)   backward {b_ms:.3f} ms ({N * 160 / b_ms / 1e9:.2f} TB/s on 160 B)")
rows, sc = kernels.ref_mix_backward(g, ref, q, t, w)
print("scalars", sc.cpu().numpy(), "row checksum", float(rows.sum()), float(kernels.ref_mix_forward(g, ref, t, w).sum()))
