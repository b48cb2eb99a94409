"""Dev helper: in-kernel s_memtime shares of the planned mode-N kernel (BEAR_DEBUG_TIMING=1)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels, _lib
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
plan = kernels.Plan(t["train"], 5)
os.environ["BEAR_DEBUG_TIMING"] = "1"
kernels.dm_prior_planned(plan, f, 0.0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); kernels.dm_prior_planned(plan, f, 0.0); e1.record(); torch.cuda.synchronize()
print("kernel ms (stamped build)", e0.elapsed_time(e1))
nb = 512
buf = np.zeros(nb * NW * 4, dtype=np.uint64)
L = _lib.lib()
L.bear_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert L.bear_debug_read_timing(plan.ws.handle, buf.ctypes.data, buf.size) == 0
a = buf.reshape(nb, NW, 4).astype(np.float64)
tiles = a[:, :, 3].mean()
print("tiles per block", tiles)
for k, nme in enumerate(["top wait (dma+barrier)", "stage issue", "work"]):
    print(f"{nme:24s} per tile: mean {a[:,:,k].mean()/tiles:8.0f}   per wave:", " ".join("%5.0f" % (a[:, w, k].mean() / tiles) for w in range(NW)))
