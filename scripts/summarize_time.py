"""Device k-mer transition counting: n random letters in reads of 150, lag 13 (emit + radix sort + run-length reduce)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bear_amd import summarize
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
lag = int(sys.argv[2]) if len(sys.argv) > 2 else 13
dev = torch.device("cuda", 0)
reads = n // 150
body = torch.randint(0, 4, (reads, 150), dtype=torch.uint8, device=dev)
text = torch.cat([torch.full((reads, 1), 5, dtype=torch.uint8, device=dev), body, torch.full((reads, 1), 4, dtype=torch.uint8, device=dev)], 1).reshape(-1)
grp = torch.zeros_like(text)
summarize.count_transitions(text[:1000000].contiguous(), grp[:1000000].contiguous(), lag, 1)
torch.cuda.synchronize(); t = time.time()
kmers, counts = summarize.count_transitions(text, grp, lag, 1)
dt = time.time() - t
print("lag %d: %d positions -> %d rows in %.3f s (%.2e transitions/s incl. download of the table)" % (lag, text.numel(), kmers.shape[0], dt, text.numel() / dt))
torch.cuda.synchronize(); t = time.time()
kd, cd = summarize.count_transitions(text, grp, lag, 1, on_device=True)
torch.cuda.synchronize(); dt = time.time() - t
print("      on the device only: %.3f s (%.2e transitions/s)" % (dt, text.numel() / dt))
