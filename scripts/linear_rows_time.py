"""Dev helper: the linear AR function as rows (bear_linear_forward_f64 / bear_linear_backward_f64), random and k-mer order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import kernels
N = int(float(os.environ.get("N", "1e8")))
LAG = int(os.environ.get("LAG", "13"))
dev = torch.device("cuda", 0)
gen = torch.Generator(dev).manual_seed(20211012)
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=gen)
mat = (0.05 * torch.randn(LAG, 5, 5, dtype=torch.float64, device=dev, generator=gen)).contiguous()
packed = kernels.pack_kmers(codes)
del codes
q = torch.randn(N, 5, dtype=torch.float64, device=dev, generator=gen)
q[torch.rand(N, device=dev, generator=gen) < 0.3] = 0      # contexts without counts


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


for name in ("random order", "k-mer order"):
    if name == "k-mer order":
        order = kernels.kmer_order(packed, LAG)
        packed, q = kernels.gather_rows(packed, order), kernels.gather_rows(q, order)
        del order
    prior = kernels.linear_forward(packed, mat, LAG)
    ms_f = timed(lambda: kernels.linear_forward(packed, mat, LAG))
    ms_b = timed(lambda: kernels.linear_backward(packed, LAG, prior, q))
    print(f"{name}: forward {ms_f:.3f} ms ({N * 48 / ms_f / 1e9:.2f} TB/s on 8 + 40 B)   backward {ms_b:.3f} ms "
          f"({N * 88 / ms_b / 1e9:.2f} TB/s on 8 + 40 + 40 B)", flush=True)
    del prior
