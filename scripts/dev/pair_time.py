import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
N, LAG = 100_000_000, 13
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
key = torch.zeros(N, dtype=torch.int64, device=dev)
for l in range(LAG):
    key = key * 6 + codes[:, l].to(torch.int64)
order = torch.argsort(key); del key
tr_s = t["train"][order].contiguous(); packed_s = kernels.linear_index(kernels.pack_kmers(codes[order].contiguous()), LAG)
torch.cuda.synchronize(); t0 = time.perf_counter()
plan = kernels.Plan(tr_s, 5)
torch.cuda.synchronize(); t1 = time.perf_counter()
plan.pair_contexts(packed_s, LAG)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(os.environ.get("BEAR_AMD_LIB", "default").split("/")[-1], f"plan {t1 - t0:.3f} s, pair_contexts {t2 - t1:.3f} s")
