"""Dev helper: the CNN training step (bear_net_cnn_train_reduce_f64: forward + DM step + backward over the plan's live lists),
contexts in k-mer order, as bench.py times it; optional BEAR_AMD_LIB selects a developer build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs
n, lag, fw = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000, 13, 8
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
plan = kernels.Plan(t, 5)
codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
key = torch.zeros(n, dtype=torch.int64, device=dev)
for l in range(lag):
    key = key * 6 + codes[:, l].to(torch.int64)
packed = kernels.pack_kmers(codes[torch.argsort(key)].contiguous()); del key, codes
if os.environ.get("CNN_STEP_KEEP"):      # as bear_net.train holds a batch: the contexts without training counts left out
    keep = (t != 0).any(dim=1).nonzero().squeeze(1)
    t, packed = t.index_select(0, keep).contiguous(), packed.index_select(0, keep).contiguous()
    plan = kernels.Plan(t, 5)
    print("kept %d of %d contexts" % (keep.numel(), n)); del keep
_, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev, generator=torch.Generator(dev).manual_seed(10))
flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
theta = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), flat]).contiguous()
bufs = kernels.cnn_step_buffers(t.shape[0], lag, fw, dev)
pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)
fn = lambda: kernels.net_cnn_train_reduce(plan, packed, lag, fw, theta, bufs, pk)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): fn()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("BEAR_AMD_LIB", "default").split("/")[-1], "training step, k-mer order: %.2f ms per %.0e contexts; sum LL %.12e" % (e0.elapsed_time(e1) / 3, n, float(pk[0])))
