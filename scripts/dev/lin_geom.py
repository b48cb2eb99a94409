"""Fused linear step (bear_dm_linear_f64) on the synthetic tables, one library per process (BEAR_AMD_LIB picks a build variant):
    python scripts/dev/lin_geom.py [table ...]      tables: replace13 (1e8 13-mers drawn with replacement: rounds 1-5's table),
                                                    distinct13 (6e7 distinct 13-mers), distinct14 (1e8 distinct 14-mers),
                                                    small13 (1e7 distinct 13-mers: configs[2]), kept13 (distinct13 without its empty rows)
Prints per table: plain / paired kernel ms, sum LL, d/dh and |d/d mat| sums (to compare libraries: same sums)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import ctypes

from bear_amd import kernels, _lib

SEED = 20211012
dev = torch.device("cuda", 0)


def timed(fn, reps=5, groups=6):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def table(kind):
    if kind == "replace13":
        n, lag = 100_000_000, 13
        codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(SEED))
        key = torch.zeros(n, dtype=torch.int64, device=dev)
        for l in range(lag):
            key = key * 4 + codes[:, l].to(torch.int64)
        codes = codes[torch.argsort(key)].contiguous()
        del key
    else:
        n, lag = {"distinct13": (60_000_000, 13), "kept13": (60_000_000, 13), "distinct14": (100_000_000, 14), "small13": (10_000_000, 13)}[kind]
        codes = kernels.synth_kmer_codes(SEED, 0, n, lag, dev, sort=True)
    train = kernels.synth_counts(SEED, 0, n, dev, want=("train",))["train"]
    if kind == "kept13":
        keep = (train != 0).any(dim=1).nonzero().squeeze(1)
        train, codes = train.index_select(0, keep).contiguous(), codes.index_select(0, keep).contiguous()
    return n, lag, train, codes


def main():
    kinds = sys.argv[1:] or ["replace13", "distinct13", "small13"]
    lib = os.environ.get("BEAR_AMD_LIB", "default").split("/")[-1]
    for kind in kinds:
        n, lag, train, codes = table(kind)
        mat = 0.05 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(10))
        packed = kernels.linear_index(kernels.pack_kmers(codes), lag)
        del codes
        plan = kernels.Plan(train, 5)
        ms_plain = timed(lambda: kernels.dm_linear(plan, packed, mat, 0.0))
        out_p = [x.clone() for x in kernels.dm_linear(plan, packed, mat, 0.0)]
        paired = plan.pair_contexts(packed, lag)
        ms_pair = timed(lambda: kernels.dm_linear(plan, packed, mat, 0.0))
        out = kernels.dm_linear(plan, packed, mat, 0.0)
        torch.cuda.synchronize()
        same = float((out[1] - out_p[1]).abs().max() / out_p[1].abs().max())
        print(f"{lib} {kind}: n={n} rows={train.shape[0]} tiles={len(plan.tiles()[0])} paired={paired} {plan.pair_info()} "
              f"plain {ms_plain:.4f} ms  paired {ms_pair:.4f} ms  per 1e8: {ms_pair * 1e8 / n:.4f}  "
              f"LL {float(out[0][0]):.15e} dh {float(out[0][1]):.15e} |dmat| {float(out[1].abs().sum()):.12e} paired-vs-plain {same:.2e}", flush=True)
        L = _lib.lib()
        if hasattr(L, "bear_dbg_lin_stamps"):       # -DBEAR_DEV_BUILD -DLIN_STAMPS: the waves' clocks per section of the tile loop
            L.bear_dbg_lin_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
            torch.cuda.synchronize()
            L.bear_dbg_lin_stamps(None, 1)
            kernels.dm_linear(plan, packed, mat, 0.0)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 8)()
            L.bear_dbg_lin_stamps(buf, 0)
            names = ["B items", "wait DMA", "barrier after B", "staging", "C", "A", "wait read-backs", "row stores + barrier"]
            n_t = len(plan.tiles()[0])
            print("    clocks per tile and wave: " + "  ".join(f"{nm} {buf[k] / (16 * n_t):.0f}" for k, nm in enumerate(names))
                  + f"  total {sum(buf) / (16 * n_t):.0f}", flush=True)
        del plan, packed, train, out, out_p
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
