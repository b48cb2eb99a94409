"""Dev helper: cost of the torch AR-function path (forward + backward through given gradient rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import ar_funcs
N = int(float(os.environ.get("N", "1e7")))
dev = torch.device("cuda", 0)
codes = torch.randint(0, 4, (N, 13), dtype=torch.int8, device=dev)
g = torch.randn(N, 5, dtype=torch.float64, device=dev)
for name, kw in [("linear", {}), ("cnn", {})]:
    torch.manual_seed(0)
    f, params = getattr(ar_funcs, "make_ar_func_" + name)(13, 4, device=dev, **kw)
    def step():
        for p in params: p.grad = None
        y = f(codes)
        y.backward(g)
    try:
        step(); torch.cuda.synchronize()
        t0 = time.time(); step(); step(); torch.cuda.synchronize()
        print(f"{name}: N={N} fwd+bwd {(time.time()-t0)/2*1e3:.1f} ms, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    except Exception as ex:
        print(name, "failed:", type(ex).__name__, str(ex)[:200])
