"""Developer: where a streamed batch's host time goes (cProfile of bear_net.train under BEAR_AMD_STREAM=1)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baseline_configs import _table
from bear_amd import ar_funcs, bear_net, dataloader
rows, batches, epochs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000, 3, 4
dev = torch.device("cuda", 0)
kmers, counts = _table(rows * batches, 13, dev, ("train",))
data = dataloader.CountDataset(kmers, counts, "dna", rows)
os.environ["BEAR_AMD_STREAM"] = "1"
run = lambda: bear_net.train(data.repeat(epochs), rows * batches, epochs, 0, "dna", 13, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False)
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
