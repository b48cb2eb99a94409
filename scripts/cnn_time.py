"""Times the fused CNN head (forward, backward) on n random contexts at lag 13."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bear_amd import ar_funcs, kernels
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
lag, fw = 13, 8
dev = torch.device("cuda", 0)
codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev)
packed = kernels.pack_kmers(codes)
_, params = ar_funcs.make_ar_func_cnn(lag, 4, device=dev)
flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
g = torch.randn(n, 5, dtype=torch.float64, device=dev)
for name, fn in (("forward", lambda: kernels.cnn_forward(packed, flat, lag, fw)),):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(3): out = fn()
    torch.cuda.synchronize(); print(name, "ms per 1e8 contexts:", (time.time() - t) / 3 * 1e3 * 1e8 / n)
prior, t1 = out
fn = lambda: kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g)
fn(); torch.cuda.synchronize(); t = time.time()
for _ in range(3): fn()
torch.cuda.synchronize(); print("backward ms per 1e8 contexts:", (time.time() - t) / 3 * 1e3 * 1e8 / n)
