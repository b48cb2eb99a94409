"""Dev helper: the convolutional training step (bear_net_cnn_train_reduce_f64) on the table bear_net.train keeps resident (k-mer
order, contexts without training counts left out), with and without prefix levels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs
N = int(float(os.environ.get("N", "1e8"))); LAG, FW = 13, 8
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))["train"]
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
key = torch.zeros(N, dtype=torch.int64, device=dev)
for l in range(LAG):
    key = key * 6 + codes[:, l].to(torch.int64)
order = torch.argsort(key); del key
keep = (t[order] != 0).any(dim=1).nonzero().squeeze(1)
tr = t[order].index_select(0, keep).contiguous(); packed = kernels.pack_kmers(codes[order].index_select(0, keep).contiguous())
del order, keep, codes, t
n = tr.shape[0]
_, params = ar_funcs.make_ar_func_cnn(LAG, 4, filter_width=FW, device=dev, generator=torch.Generator(dev).manual_seed(10))
flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
theta = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), flat]).contiguous()
plan = kernels.Plan(tr, 5)
bufs = kernels.cnn_step_buffers(n, LAG, FW, dev)
pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


ms0 = timed(lambda: kernels.net_cnn_train_reduce(plan, packed, LAG, FW, theta, bufs, pk))
ref = pk.clone()
print(f"{n} kept contexts of {N}; plain step {ms0:.2f} ms")
print("levels attached:", plan.attach_cnn_levels(packed, LAG, FW), "plan bytes per context", plan.nbytes / n)
ms1 = timed(lambda: kernels.net_cnn_train_reduce(plan, packed, LAG, FW, theta, bufs, pk))
print(f"with prefix levels {ms1:.2f} ms; ELBO rel diff {abs((pk[0]-ref[0])/ref[0]).item():.2e}, max grad diff {(pk[2:]-ref[2:]).abs().max().item():.3e} of {ref[2:].abs().max().item():.3e}")
print("level rows", plan.cnn_level_rows(with_letters=True), "window tables (position, windows)", plan.cnn_window_rows())
os.environ["BEAR_AMD_CNN_NO_WINDOWS"] = "1"
plan.attach_cnn_levels(packed, LAG, FW)
ms2 = timed(lambda: kernels.net_cnn_train_reduce(plan, packed, LAG, FW, theta, bufs, pk))
print(f"prefix levels without window tables {ms2:.2f} ms; max grad diff {(pk[2:]-ref[2:]).abs().max().item():.3e}")
