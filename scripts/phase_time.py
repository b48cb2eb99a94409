"""Dev helper: time mode-N sorted kernel with phase cut-offs (BEAR_DEBUG_STOP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import kernels
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
f = kernels.synth_prior(20211012, 0, N, dev)
for stop in sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "0"):
    os.environ["BEAR_DEBUG_STOP"] = stop
    kernels.dm_prior(t["train"], f, 0.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): kernels.dm_prior(t["train"], f, 0.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"stop={stop} {ms:8.3f} ms  {N*60/ms/1e6:8.1f} GB/s")
