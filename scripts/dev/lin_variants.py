import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs
N, LAG = 100_000_000, 13
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
torch.manual_seed(0)
f, (mat,) = ar_funcs.make_ar_func_linear(LAG, 4, device=dev)
key = torch.zeros(N, dtype=torch.int64, device=dev)
for l in range(LAG):
    key = key * 6 + codes[:, l].to(torch.int64)
order = torch.argsort(key); del key
tr_s = t["train"][order].contiguous(); packed_s = kernels.linear_index(kernels.pack_kmers(codes[order].contiguous()), LAG); del order
plan_s = kernels.Plan(tr_s, 5)
if os.environ.get("PAIRED"):
    plan_s.pair_contexts(packed_s, LAG)
fn = lambda: kernels.dm_linear(plan_s, packed_s, mat.detach(), 0.0)
for _ in range(30): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fn()
e1.record(); torch.cuda.synchronize()
best = e0.elapsed_time(e1) / 5
for _ in range(5):
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(os.environ.get("BEAR_AMD_LIB", "default").split("/")[-1], "sorted%s: %.3f ms" % (" paired" if os.environ.get("PAIRED") else "", best))
