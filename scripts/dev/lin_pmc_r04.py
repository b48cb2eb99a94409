"""Dev helper for scripts/dev/lin_pmc_r04.sh: the fused linear step on a k-mer-sorted 1e8-context table, four launches with the
plain lists (dm_linear_plan_kernel<false, false>) and four with paired lists (<false, true>): the two forms carry different kernel
names, so rocprofv3's per-kernel counters separate them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs
N, LAG = int(float(os.environ.get("N", "1e8"))), 13
dev = torch.device("cuda", 0)
if os.environ.get("TABLE") == "distinct13":      # round 6: DISTINCT contexts (kernels.synth_kmer_ids); N <= 4^13
    N = min(N, 60_000_000)
    tr = kernels.synth_counts(20211012, 0, N, dev, want=("train",))["train"]
    idx = kernels.linear_index(kernels.pack_kmers(kernels.synth_kmer_codes(20211012, 0, N, LAG, dev, sort=True)), LAG)
else:
    t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))["train"]
    codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
    key = torch.zeros(N, dtype=torch.int64, device=dev)
    for l in range(LAG):
        key = key * 6 + codes[:, l].to(torch.int64)
    order = torch.argsort(key); del key
    tr = t[order].contiguous(); idx = kernels.linear_index(kernels.pack_kmers(codes[order].contiguous()), LAG); del order, codes, t
torch.manual_seed(0)
_, (mat,) = ar_funcs.make_ar_func_linear(LAG, 4, device=dev)
plan = kernels.Plan(tr, 5)
for _ in range(4):
    kernels.dm_linear(plan, idx, mat.detach(), 0.0)
assert plan.pair_contexts(idx, LAG)
for _ in range(4):
    kernels.dm_linear(plan, idx, mat.detach(), 0.0)
torch.cuda.synchronize()
print("contexts", N, "live", int((tr != 0).any(dim=1).sum()))
