"""Developer: per-wave phase shares of eval_plan_kernel from a -DEVP_STAMPS build (BEAR_AMD_LIB=build_variants/libbear_evpstamps.so)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels, _lib
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "test"))
f = kernels.synth_prior(20211012, 0, n, dev)
if os.environ.get("EVP_ALL_ROWS"):
    plan, ids = kernels.EvalPlan(t["test"], t["train"]), None
else:       # as evaluation() holds a batch: the contexts with held-out counts only
    keep = (t["test"] != 0).any(dim=1).nonzero().squeeze(1)
    f = f.index_select(0, keep).contiguous()
    plan, ids = kernels.EvalPlan(t["test"].index_select(0, keep).contiguous(), t["train"].index_select(0, keep).contiguous()), keep.to(torch.int32)
L = _lib.lib()
L.bear_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
names = ["wait for tile", "H units", "cell units", "V units", "tie units", "tickets/rest", "DMA issue", "DMA wait"]
for label, hs, van in (("1h+AR+3van", [1.0], [0.1, 1.0, 10.0]),):
    kernels.evaluate_planned(plan, f, hs, van, row_ids=ids); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); kernels.evaluate_planned(plan, f, hs, van, row_ids=ids); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nb, nw = 256, 16          # EVP_THREADS 1024: 14 compute waves + 2 DMA waves
    buf = np.zeros(nb * nw * 8, dtype=np.uint64)
    assert L.bear_debug_read_timing(plan.ws.handle, buf.ctypes.data, buf.size) == 0
    a = buf.reshape(nb, nw, 8).astype(np.float64)
    for role, sl in (("compute waves", slice(0, 14)), ("DMA waves", slice(14, 16))):
        tot = a[:, sl].sum(-1).mean()
        print(f"{label} [{role}]: kernel {ms:.3f} ms (stamped build); ticks per wave {tot:.0f}")
        for k, nme in enumerate(names):
            print(f"  {nme:14s} {a[:, sl, k].mean() / tot * 100:5.1f} %   ({a[:, sl, k].mean() / tot * ms * 1e3:7.1f} us)")
