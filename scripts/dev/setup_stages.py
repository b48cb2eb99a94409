"""Dev helper: where the set-up time of bear_net.train / bear_ref.evaluation goes on a 1e7-row table (synchronised wall clock around
the internal steps: ResidentBatches, plans, k-mer order, packing, graph capture, the steps themselves)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bear_amd import _train, ar_funcs, bear_net, bear_ref, dataloader, kernels
N = int(float(os.environ.get("N", "1e7")))
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, N, dev)
codes = torch.randint(0, 4, (N, 13), dtype=torch.int64, device=dev, generator=torch.Generator(dev).manual_seed(1))
kmers = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[codes].cpu().numpy()
counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in ("train", "test", "ref")])
del t, codes
data = dataloader.CountDataset(kmers, counts, "dna", N) if hasattr(dataloader, "CountDataset") else None
which = sys.argv[1] if len(sys.argv) > 1 else "net_train"


def run():
    if which == "net_train":
        bear_net.train(data.repeat(50), N, 50, 0, "dna", 13, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False)
    elif which == "ref_eval":
        f, params = bear_ref._make_ref_ar_func(13, 4, ar_funcs.make_ar_func_stop, {}, device=dev)
        bear_ref.evaluation(data, 0, 1, 2, "dna", torch.tensor(0.8), f, np.array([0.1, 1.0, 10.0]))
    elif which == "net_eval":
        f, params = ar_funcs.make_ar_func_linear(13, 4, device=dev)
        bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.8), f, np.array([0.1, 1.0, 10.0]))
    torch.cuda.synchronize()


os.environ["AMD_SERIALIZE_KERNEL"] = "3"      # launches synchronous: host time = device time under the profiler
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); run(); pr.disable()
    print(f"== {which} pass {rep}: {time.perf_counter() - t0:.3f} s")
    if rep == 1 or True:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
