"""Dev helper: the fused linear step's launch time against the table size (k-mer order, paired lists): slope = per-context cost, intercept = what a launch costs whatever its size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, _lib
import ctypes
LAG = 13
dev = torch.device("cuda", 0)
mat = 0.05 * torch.randn(LAG, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(10))
SIZES = [int(float(a)) for a in sys.argv[1:]] or [300_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000]
for n in SIZES:
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
    codes = torch.randint(0, 4, (n, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
    fixed = 0
    while 4 ** (LAG - fixed - 1) * 1.49 >= n:      # as dense in k-mer space as the 1e8 benchmark table
        fixed += 1
    codes[:, :fixed] = 0
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(LAG):
        key = key * 6 + codes[:, l].to(torch.int64)
    order = torch.argsort(key)
    tr, cd = t[order].contiguous(), codes[order].contiguous()
    keep = (tr != 0).any(dim=1).nonzero().squeeze(1)
    tr, cd = tr.index_select(0, keep).contiguous(), cd.index_select(0, keep).contiguous()
    idx = kernels.linear_index(kernels.pack_kmers(cd), LAG)
    plan = kernels.Plan(tr, 5)
    plan.pair_contexts(idx, LAG)
    fn = lambda: kernels.dm_linear(plan, idx, mat, 0.0)
    for _ in range(50): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{n:>9d} contexts ({tr.shape[0]} kept, {len(plan.tiles()[0])} tiles): {best * 1e3:8.1f} us per launch", flush=True)
    L = _lib.lib()
    if hasattr(L, "bear_dbg_lin_pe_stamps"):       # -DBEAR_DEV_BUILD -DLIN_STAMPS: sections of block 0's prologue / epilogue, clocks
        buf = (ctypes.c_ulonglong * 12)()
        L.bear_dbg_lin_pe_stamps(buf)
        names = ["params+tables0", "group tables", "first desc", "first tile", "first A", "tile loop", "overflow+hist", "fold+accum", "block sums",
                 "last arrival", "take accum", "finalize+apply"]
        print("      " + "  ".join(f"{nm} {buf[k]}" for k, nm in enumerate(names)), flush=True)
