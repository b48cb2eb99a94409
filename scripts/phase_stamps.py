"""Dev helper: in-kernel s_memtime phase shares of the mode-N sorted kernel (BEAR_DEBUG_STOP=9)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels, _lib
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
ws = kernels.default_workspace(dev)
os.environ["BEAR_DEBUG_STOP"] = "9"
kernels.dm_prior(t["train"], f, 0.0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); kernels.dm_prior(t["train"], f, 0.0); e1.record(); torch.cuda.synchronize()
print("kernel ms (stamped build)", e0.elapsed_time(e1))
nb = 512
buf = np.zeros(nb * 8 * 6, dtype=np.uint64)
L = _lib.lib()
L.bear_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
st = L.bear_debug_read_timing(ws.handle, buf.ctypes.data, buf.size); assert st == 0
a = buf.reshape(nb, 8, 6).astype(np.float64)
names = ["wait_dma+top_barrier", "A count", "B scan", "C scatter", "D light(+idle)", "D heavy"]
tot = a.sum(-1)
print("per-wave total cycles: mean %.3e  (x100MHz ticks?)" % tot.mean())
for k, nme in enumerate(names):
    print(f"{nme:24s} mean {a[:,:,k].mean():12.0f}  share {a[:,:,k].sum()/tot.sum()*100:5.1f}%   wave0 {a[:,0,k].mean():12.0f}  wave7 {a[:,7,k].mean():12.0f}")
tiles = N / 512 / nb
print("tiles per block %.1f ; cycles per tile %.0f" % (tiles, tot.mean() / tiles))
