"""Developer: bank-pair collisions of the linear step's triple adds, counted from the plan's paired lists (bear_debug_pair_lists).
An LDS fp64 atomic runs as four passes of 16 lanes; a pass takes as long as its fullest bank pair (triple row mod 16) holds lanes.
Prints the mean of sum-over-passes(max lanes on a bank pair) per instruction (4 = conflict-free)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import kernels, _lib
N, LAG = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000, 13
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))
codes = torch.randint(0, 4, (N, LAG), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(20211012))
key = torch.zeros(N, dtype=torch.int64, device=dev)
for l in range(LAG):
    key = key * 6 + codes[:, l].to(torch.int64)
order = torch.argsort(key); del key
tr_s = t["train"][order].contiguous(); packed_s = kernels.linear_index(kernels.pack_kmers(codes[order].contiguous()), LAG)
plan = kernels.Plan(tr_s, 5)
plan.pair_contexts(packed_s, LAG)
L = _lib.lib()
L.bear_debug_pair_lists.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
STRIDE = L.bear_debug_pair_lists(None, 0, 0, None, None)
n_t = 400
lists = np.zeros((n_t, STRIDE), dtype=np.uint16); row0 = np.zeros(n_t, dtype=np.uint64)
assert L.bear_debug_pair_lists(plan._h, 1000, n_t, lists.ctypes.data, row0.ctypes.data) == 0
words = packed_s.cpu().numpy().view(np.uint64)
npair = (LAG - 3 + 1) // 2
tot, cnt, ents, ctx = 0, 0, 0, 0
for k in range(n_t):
    m = int(lists[k, 0]); e = lists[k, 2:2 + m].astype(np.int64)
    ents += m; ctx += int((e != 0xffff).sum())
    w = words[int(row0[k]) + np.where(e == 0xffff, 0, e)]
    cl = ((w >> np.uint64(6 * npair)) & np.uint64(255)).astype(np.int64) % 16
    cl = np.where(e == 0xffff, -1, cl)
    tw = (e[1::2] != 0xffff) & (w[1::2] == w[0::2])          # two copies of one k-mer in a lane: one add
    cl[1::2] = np.where(tw, -1, cl[1::2])
    pad = (-m) % 128
    cl = np.concatenate([cl, -np.ones(pad, dtype=np.int64)]).reshape(-1, 64, 2)
    for slot in (0, 1):
        q = cl[:, :, slot].reshape(-1, 4, 16)
        mx = np.zeros(q.shape[:2], dtype=np.int64)
        for v in range(16):
            mx = np.maximum(mx, (q == v).sum(-1))
        live = (q >= 0).any(-1).any(-1)
        tot += mx.sum(); cnt += int(live.sum())
print(f"tiles {n_t}: contexts {ctx}, entries {ents} ({ents / ctx:.3f} per context), passes-sum of max lanes per bank pair: {tot / cnt:.2f} per instruction (4 = none)")
