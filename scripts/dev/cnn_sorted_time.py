"""Dev helper: cnn forward / backward on contexts in random order and in k-mer order (shared windows), results compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs
n, lag, fw = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000, 13, 8
dev = torch.device("cuda", 0)
gen = torch.Generator(dev).manual_seed(3)
codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
key = torch.zeros(n, dtype=torch.int64, device=dev)
for l in range(lag):
    key = key * 6 + codes[:, l].to(torch.int64)
order = torch.argsort(key)
torch.manual_seed(5)
_, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev)
flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {}
for name, cd, tr in (("random order", codes, t), ("k-mer order", codes[order].contiguous(), t[order].contiguous())):
    packed = kernels.pack_kmers(cd)
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    _, g = kernels.dm_prior_planned(kernels.Plan(tr, 5), prior, 0.0, want_grad=True)
    grad = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g)
    f_ms = timed(lambda: kernels.cnn_forward(packed, flat, lag, fw))
    b_ms = timed(lambda: kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g))
    print("%-13s forward %.2f ms, backward %.2f ms per %.0e contexts" % (name, f_ms, b_ms, n))
    res[name] = (prior, grad)
back = torch.empty_like(res["k-mer order"][0]); back[order] = res["k-mer order"][0]
print("prior rows, sorted vs random order: max |diff| %.3e" % float((back - res["random order"][0]).abs().max()))
print("gradients: max |diff| / max |grad| %.3e" % float((res["k-mer order"][1] - res["random order"][1]).abs().max() / res["random order"][1].abs().max()))
