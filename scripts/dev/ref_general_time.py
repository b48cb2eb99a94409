"""Dev helper: bear_ref with a parametrised net function -- cost of the mixing (nw net + jukes_cantor(ref, tau)) / (nw + 1)
around the net function, forward + backward through given gradient rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import ar_funcs, bear_ref, kernels
N = int(float(os.environ.get("N", "1e7")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
codes = torch.randint(0, 4, (N, 13), dtype=torch.int8, device=dev)
ref_in = bear_ref._ref_input(t["ref"])
g = torch.randn(N, 5, dtype=torch.float64, device=dev)


def timed(step, reps=3):
    step(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3


for name in ("linear", "cnn"):
    torch.manual_seed(0)
    make = getattr(ar_funcs, "make_ar_func_" + name)
    f, params = bear_ref._make_ref_ar_func(13, 4, make, {}, device=dev)
    net, net_params = make(13, 4, device=dev)

    def mixed():
        for p in params: p.grad = None
        f(codes, ref_in).backward(g)

    def bare():
        for p in net_params: p.grad = None
        net(codes).backward(g)
    print(f"{name}: N={N:.0e}  net alone {timed(bare):.1f} ms   net + reference mixing {timed(mixed):.1f} ms", flush=True)
