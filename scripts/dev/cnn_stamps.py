"""Dev helper: phase clocks of cnn_backward_kernel (library built with -DCNN_STAMPS, named by BEAR_AMD_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs, _lib
n, lag, fw = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000, 13, 8   # argv[2] = "sorted": contexts in k-mer order
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev)
if len(sys.argv) > 2 and sys.argv[2] == "sorted":
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        key = key * 6 + codes[:, l].to(torch.int64)
    order = torch.argsort(key); del key
    codes, t = codes[order].contiguous(), t[order].contiguous(); del order
packed = kernels.pack_kmers(codes); del codes
_, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev)
flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
_, g = kernels.dm_prior_planned(kernels.Plan(t, 5), prior, 0.0, want_grad=True)
kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g); torch.cuda.synchronize()
L = _lib.lib()
L.bear_dbg_cnn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.bear_dbg_cnn_stamps(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g); e1.record(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
L.bear_dbg_cnn_stamps(buf, 0)
tot = sum(buf)
names = ["conv+norm+elu+stage", "MFMA d weights1", "MFMA d e0 + handback", "dy/dn + 2 column sums", "norm backward + stage",
         "MFMA d filters (+ one-hot)", "per-tile head (loads, layer 1, small sums)", "positions done per distinct window (shared windows)"]
print("backward %.2f ms for %.0e contexts" % (e0.elapsed_time(e1), n))
for k in range(8):
    print("%-45s %5.1f %%" % (names[k], 100.0 * buf[k] / tot))
