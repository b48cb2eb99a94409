"""Statistics of the synthetic k=13 table that size the evaluation kernel: active rows, cells, ties (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import kernels

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda")
t = kernels.synth_counts(20211012, 0, n, dev)
tr, te, rf = (t[k].long() for k in ("train", "test", "ref"))
for name, x in (("train", tr), ("test", te), ("ref", rf)):
    tot = x.sum(1)
    print(f"{name}: rows with n>0 {float((tot > 0).float().mean()):.4f}  cells>0 per row {float((x > 0).float().sum(1).mean()):.4f}  "
          f"mean n {float(tot.float().mean()):.3f}  max c {int(x.max())}  cells with c>24 per row {float((x > 24).float().sum(1).mean()):.5f}  "
          f"rows n>24 {float((tot > 24).float().mean()):.5f}")
    if name == "test":
        hist = torch.bincount(x[x > 0].clamp(max=30))
        print("  cell count histogram (1..30+):", (hist[1:].float() / hist[1:].sum()).cpu().numpy().round(4).tolist())
        hist = torch.bincount(tot[tot > 0].clamp(max=30))
        print("  row total histogram (1..30+):", (hist[1:].float() / hist[1:].sum()).cpu().numpy().round(4).tolist())
active = te.sum(1) > 0
top = tr.max(1).values
ntop = (tr == top[:, None]).sum(1)
print(f"active rows {float(active.float().mean()):.4f}; of them vanilla arg-max tied (>=2 letters at the top train count): "
      f"{float(((ntop >= 2) & active).float().sum() / active.float().sum()):.4f}; mean contenders when tied "
      f"{float(ntop[(ntop >= 2) & active].float().mean()):.3f}; all-zero train rows among active {float(((top == 0) & active).float().sum() / active.float().sum()):.4f}")
print("train row total histogram among active rows (0..40+):",
      (torch.bincount(tr.sum(1)[active].clamp(max=40)).float() / active.float().sum()).cpu().numpy().round(4).tolist())
print("train cell value histogram among active rows (0..40+):",
      (torch.bincount(tr[active].reshape(-1).clamp(max=40)).float() / (5 * active.float().sum())).cpu().numpy().round(4).tolist())
