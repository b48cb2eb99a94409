"""Dev helper: the fused linear-head training step (bear_dm_linear_f64) on random-order and on k-mer-sorted contexts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bear_amd import kernels, ar_funcs
N = int(float(os.environ.get("N", "1e8")))
LAG = int(os.environ.get("LAG", "13"))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
torch.manual_seed(0)
f, (mat,) = ar_funcs.make_ar_func_linear(LAG, 4, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


packed = kernels.linear_index(kernels.pack_kmers(codes), LAG)
plan = kernels.Plan(t["train"], 5)
ref_out, ref_g = kernels.dm_linear(plan, packed, mat.detach(), 0.0)
for ar in (False, True):
    ms = timed(lambda: kernels.dm_linear(plan, packed, mat.detach(), 0.0, train_ar=ar))
    print(f"random order  ar={ar}: {ms:.3f} ms ({N / ms / 1e6:.2f} Gctx/s)")
# the same table with its rows sorted by k-mer (what bear_net.train does at upload)
key = torch.zeros(N, dtype=torch.int64, device=dev)
for l in range(LAG):
    key = key * 6 + codes[:, l].to(torch.int64)
order = torch.argsort(key)
del key
tr_s = t["train"][order].contiguous()
packed_s = kernels.linear_index(kernels.pack_kmers(codes[order].contiguous()), LAG)
del order
plan_s = kernels.Plan(tr_s, 5)
out_s, g_s = kernels.dm_linear(plan_s, packed_s, mat.detach(), 0.0)
print("sorted vs random order: ELBO rel diff %.2e, d/dh rel diff %.2e, max |d mat| diff %.2e of %.2e" % (
    abs((out_s[0] - ref_out[0]) / ref_out[0]).item(), abs((out_s[1] - ref_out[1]) / ref_out[1]).item(),
    (g_s - ref_g).abs().max().item(), ref_g.abs().max().item()))
for ar in (False, True):
    ms = timed(lambda: kernels.dm_linear(plan_s, packed_s, mat.detach(), 0.0, train_ar=ar))
    print(f"k-mer sorted  ar={ar}: {ms:.3f} ms ({N / ms / 1e6:.2f} Gctx/s)")
# ... and the paired form (bear_plan_pair_contexts): neighbours with equal leading letters two at a time
ok = plan_s.pair_contexts(packed_s, LAG)
out_p, g_p = kernels.dm_linear(plan_s, packed_s, mat.detach(), 0.0)
print("paired:", ok, "vs plain: ELBO rel diff %.2e, max |d mat| diff %.2e of %.2e" % (
    abs((out_p[0] - out_s[0]) / out_s[0]).item(), (g_p - g_s).abs().max().item(), g_s.abs().max().item()))
for ar in (False, True):
    ms = timed(lambda: kernels.dm_linear(plan_s, packed_s, mat.detach(), 0.0, train_ar=ar))
    print(f"k-mer sorted, paired  ar={ar}: {ms:.3f} ms ({N / ms / 1e6:.2f} Gctx/s)")
# as bear_net.train keeps the batch: the contexts without training counts left out
keep = (tr_s != 0).any(dim=1).nonzero().squeeze(1)
tr_k, pk_k = tr_s.index_select(0, keep).contiguous(), packed_s.index_select(0, keep).contiguous()
plan_k = kernels.Plan(tr_k, 5)
ms = timed(lambda: kernels.dm_linear(plan_k, pk_k, mat.detach(), 0.0))
print(f"kept rows only, plain: {ms:.3f} ms")
print("paired:", plan_k.pair_contexts(pk_k, LAG))
ms = timed(lambda: kernels.dm_linear(plan_k, pk_k, mat.detach(), 0.0))
print(f"kept rows only, paired: {ms:.3f} ms")
