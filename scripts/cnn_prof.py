"""Dev helper for counter passes: a few launches of the CNN kernels at 1e7 contexts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import ar_funcs, kernels
n, lag, fw = 10_000_000, 13, 8
dev = torch.device("cuda", 0)
packed = kernels.pack_kmers(torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev))
_, params = ar_funcs.make_ar_func_cnn(lag, 4, device=dev)
flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
g = torch.randn(n, 5, dtype=torch.float64, device=dev)
for _ in range(2):
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g)
torch.cuda.synchronize()
