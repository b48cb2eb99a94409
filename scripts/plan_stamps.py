"""Dev helper: per-wave phase shares of dm_prior_plan_kernel from a -DPLN_STAMPS build
(BEAR_AMD_LIB=build_variants/stamps.so).  Slots: 0 prologue, 1 own-DMA wait, 2 barrier, 3 DMA issue,
4 first ticket draw, 5 large-count work, 6 item units, 7 context chunks."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels, _lib
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
plan = kernels.Plan(t["train"], 5)
L = _lib.lib()
L.bear_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
names = ["prologue", "own-DMA wait", "barrier", "DMA issue", "first ticket", "large-count", "item units", "ctx chunks"]
for label, kw in [("net", {}), ("net_norm", {"normalized": True})]:
    kernels.dm_prior_planned(plan, f, 0.0, **kw); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); kernels.dm_prior_planned(plan, f, 0.0, **kw); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nb, nw = 256, 16
    buf = np.zeros(nb * nw * 8, dtype=np.uint64)
    assert L.bear_debug_read_timing(plan.ws.handle, buf.ctypes.data, buf.size) == 0
    a = buf.reshape(nb, nw, 8).astype(np.float64)
    tot = a.sum(-1).mean()
    print(f"{label}: kernel {ms:.3f} ms (stamped build); ticks per wave {tot:.0f} -> {tot / ms / 1e3:.1f} ticks/us")
    n_dma = int(os.environ.get("DMA_WAVES", "2"))      # the block's last waves only stream tiles into LDS
    for k, nme in enumerate(names):
        c, d = a[:, :nw - n_dma, k].mean(), a[:, nw - n_dma:, k].mean()
        print(f"  {nme:14s} all waves {a[:, :, k].mean() / tot * 100:5.1f} %   compute waves {c / tot * ms * 1e3:7.1f} us   DMA waves {d / tot * ms * 1e3:7.1f} us")
