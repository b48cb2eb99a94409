"""One bear_net training step with the CNN AR function at lag 13 on n synthetic contexts:
fused forward + planned DM kernel with gradient rows + fused backward, against the torch formulation."""
import sys, time
import torch
sys.path.insert(0, ".")
from bear_amd import ar_funcs, kernels
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
torch_too = len(sys.argv) > 2
lag = 13
dev = torch.device("cuda", 0)
counts = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
plan = kernels.Plan(counts, 5)
codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev)
ar_func, params = ar_funcs.make_ar_func_cnn(lag, 4, device=dev)

def step():
    for p in params: p.grad = None
    prior = ar_func(codes)
    out, g = kernels.dm_prior_planned(plan, prior.detach().contiguous(), 0.0, want_grad=True)
    prior.backward(g)
    return out
step(); torch.cuda.synchronize(); t = time.time()
for _ in range(3): out = step()
torch.cuda.synchronize(); dt = (time.time() - t) / 3
print("fused step: %.1f ms for %.0e contexts = %.1f ms per 1e8 ; ELBO %.6e" % (dt * 1e3, n, dt * 1e3 * 1e8 / n, out[0].item()))
if torch_too:
    m = min(n, 1_000_000)
    oh_codes = codes[:m]
    ar_func.__closure__  # torch path: force by passing one-hot
    from bear_amd import core
    oh = torch.zeros((m, lag, 5), dtype=torch.float64, device=dev).scatter_(-1, oh_codes.long().unsqueeze(-1), 1.0)
    pl = kernels.Plan(counts[:m].contiguous(), 5)
    def tstep():
        for p in params: p.grad = None
        prior = ar_func(oh)
        out, g = kernels.dm_prior_planned(pl, prior.detach().contiguous(), 0.0, want_grad=True)
        prior.backward(g)
    tstep(); torch.cuda.synchronize(); t = time.time()
    tstep(); torch.cuda.synchronize(); dt = time.time() - t
    print("torch one-hot step: %.1f ms for %.0e contexts = %.0f ms per 1e8" % (dt * 1e3, m, dt * 1e3 * 1e8 / m))
