import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bear_amd import _lib, _train, ar_funcs, bear_net, bear_ref, dataloader, kernels
print("det build:", _lib.lib().bear_deterministic_build(), _lib.LIB_PATH)
dev = torch.device("cuda", 0)
path = os.path.join(os.path.dirname(_lib.__file__), "data", "ysd1_lag_5_file_0_preshuf.tsv")
data = dataloader.dataloader(path, "dna", 1500, 3)
for name, mod, make, kw, extra in (("ref_stop", bear_ref, ar_funcs.make_ar_func_stop, {}, (2,)),
                                   ("linear", bear_net, ar_funcs.make_ar_func_linear, {}, ()),
                                   ("cnn", bear_net, ar_funcs.make_ar_func_cnn, {"filter_width": 3}, ())):
    for steps in (1, 2, 3, 50):
        runs = []
        for _ in range(2):
            torch.manual_seed(10)
            loss = []
            params, h, _ = mod.train(data.repeat(steps), 1365, steps, 0, *extra, "dna", 5, make, kw, 0.01, "Adam", False, loss_save=loss)
            runs.append((np.array(loss), torch.cat([p.detach().reshape(-1) for p in params] + [h.detach().reshape(-1)]).cpu().numpy(), dict(_train.LAST_RUN)))
        same_loss = np.array_equal(runs[0][0], runs[1][0])
        same_par = np.array_equal(runs[0][1], runs[1][1])
        first_bad = int(np.argmax(runs[0][0] != runs[1][0])) if not same_loss else -1
        print(name, steps, "loss equal", same_loss, "params equal", same_par, "first differing step", first_bad,
              "graph", runs[0][2].get("graph"), "max par diff", float(np.abs(runs[0][1] - runs[1][1]).max()))
# the streaming mode-R kernel
n = 700_000
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "ref"))
pl = kernels.Plan(t["train"], 4)
a = [kernels.dm_ref_planned(pl, t["ref"], 0.1, -3.4, -4.6).clone() for _ in range(4)]
print("mode_R streaming", [torch.equal(a[0], x) for x in a], (a[0] - a[1]).tolist())
