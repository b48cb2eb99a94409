"""Dev helper for scripts/profile_round.sh: the convolutional kernels in ONE row order per process, so that rocprofv3's per-kernel
counters (which cannot tell two launches of the same kernel apart) belong to that order.
    python cnn_order_pmc.py random|sorted|levels   (N contexts via env N, default 2e7)
random / sorted: bear_cnn_forward_f64 + bear_cnn_backward_f64 over all rows; levels: the training step on the sorted table
bear_net.train keeps, prefix levels attached (cnn_forward_kernel / cnn_backward_parts_kernel once per level)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels, ar_funcs
mode = sys.argv[1]
N, LAG, FW = int(float(os.environ.get("N", "2e7"))), 13, 8
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))["train"]
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
if mode != "random":
    # as dense in k-mer space as the 1e8-context benchmark: the same number of contexts per 13-mer
    fixed = 0
    while 4 ** (LAG - fixed - 1) * 1.49 >= N:
        fixed += 1
    codes[:, :fixed] = 0
    key = torch.zeros(N, dtype=torch.int64, device=dev)
    for l in range(LAG):
        key = key * 6 + codes[:, l].to(torch.int64)
    order = torch.argsort(key)
    codes, t = codes[order].contiguous(), t[order].contiguous()
_, params = ar_funcs.make_ar_func_cnn(LAG, 4, filter_width=FW, device=dev, generator=torch.Generator(dev).manual_seed(10))
flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
if mode == "levels":
    keep = (t != 0).any(dim=1).nonzero().squeeze(1)
    tr, packed = t.index_select(0, keep).contiguous(), kernels.pack_kmers(codes.index_select(0, keep).contiguous())
    plan = kernels.Plan(tr, 5)
    print("levels", plan.attach_cnn_levels(packed, LAG, FW), plan.cnn_level_rows())
    theta = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), flat]).contiguous()
    bufs = kernels.cnn_step_buffers(tr.shape[0], LAG, FW, dev)
    pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)
    for _ in range(3):
        kernels.net_cnn_train_reduce(plan, packed, LAG, FW, theta, bufs, pk)
else:
    packed = kernels.pack_kmers(codes)
    plan = kernels.Plan(t, 5)
    for _ in range(3):
        prior, t1 = kernels.cnn_forward(packed, flat, LAG, FW)
        _, g = kernels.dm_prior_planned(plan, prior, 0.0, want_grad=True)
        kernels.cnn_backward(packed, flat, LAG, FW, t1, prior, g)
torch.cuda.synchronize()
print(mode, "done")
