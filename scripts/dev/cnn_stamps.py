"""Dev helper: phase clocks of cnn_backward_kernel (library built with -DCNN_STAMPS, named by BEAR_AMD_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs, _lib
n, lag, fw = 10_000_000, 13, 8
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
packed = kernels.pack_kmers(torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev))
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
         "MFMA d filters (+ one-hot)", "per-tile head (loads, layer 1, small sums)", "-"]
print("backward %.2f ms for %.0e contexts" % (e0.elapsed_time(e1), n))
for k in range(7):
    print("%-45s %5.1f %%" % (names[k], 100.0 * buf[k] / tot))
