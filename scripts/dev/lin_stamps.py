"""Dev helper: where the waves of dm_linear_plan_kernel spend their clocks (library built with -DLIN_STAMPS, BEAR_AMD_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs, _lib
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
tr_s = t["train"][order].contiguous(); idx = kernels.linear_index(kernels.pack_kmers(codes[order].contiguous()), LAG); del order
plan = kernels.Plan(tr_s, 5)
if os.environ.get("PAIRED"):
    print("paired:", plan.pair_contexts(idx, LAG))
for _ in range(3): kernels.dm_linear(plan, idx, mat.detach(), 0.0)
torch.cuda.synchronize()
L = _lib.lib()
L.bear_dbg_lin_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.bear_dbg_lin_stamps(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); kernels.dm_linear(plan, idx, mat.detach(), 0.0); e1.record(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
L.bear_dbg_lin_stamps(buf, 0)
tot = sum(buf)
names = ["B: items (tickets)", "wait for the DMA of the next tile", "barrier after the items", "staging the tile after next",
         "C: read-back, sums, gradient adds", "A: softmax of the next tile", "wait for the other waves' read-backs", "row stores + barrier"]
print("kernel %.3f ms" % e0.elapsed_time(e1))
for k in range(8):
    print("%-42s %5.1f %%" % (names[k], 100.0 * buf[k] / tot))
