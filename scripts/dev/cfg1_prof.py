"""Dev helper: configs[1] (bear_ref, stop prior, 1e7 contexts) through bear_ref.train under rocprofv3 --kernel-trace: what a replayed step is made of."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import baseline_configs as bc
from bear_amd import ar_funcs, bear_ref, bear_net, dataloader
dev = torch.device("cuda", 0)
n = int(float(os.environ.get("N", "1e7")))
kmers, counts = bc._table(n, 13, dev, ("train", "test", "ref"))
data = dataloader.CountDataset(kmers, counts, "dna", n)
which = os.environ.get("WHICH", "ref")
if which == "ref":
    _, e = bc._train_config(bear_ref, data, n, 13, ar_funcs.make_ar_func_stop, {}, 1000, extra=(2,))
else:
    _, e = bc._train_config(bear_net, data, n, 13, ar_funcs.make_ar_func_linear, {}, 400)
print(e)
