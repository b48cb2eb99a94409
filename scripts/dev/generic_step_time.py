"""Dev helper: microseconds per optimizer step of the GENERIC plugin loop (_train.run_autograd_steps) on BASELINE configs[0]
(the bundled 1365-row table, one batch per epoch) with an AR function of torch ops -- make_ar_func_cnn(num_filters=20), a shape
the fused kernels do not take -- eager (BEAR_AMD_NO_GRAPH=1) against captured and replayed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import _train, ar_funcs, bear_net, dataloader
YSD1 = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ysd1_lag_5_file_0_preshuf.tsv")
STEPS = int(os.environ.get("STEPS", "2000"))
data = dataloader.dataloader(YSD1, "dna", 1500, 3)
KW = {"num_filters": 20, "filter_width": 3, "kmer_layer1_width": 16}
MODES = os.environ.get("MODES", "eager,graph").split(",")
for acc in (1, 2) if len(MODES) == 2 else (1,):
    res = {}
    for mode in MODES:
        if mode == "eager":
            os.environ["BEAR_AMD_NO_GRAPH"] = "1"
        else:
            os.environ.pop("BEAR_AMD_NO_GRAPH", None)
        for rep in range(2):      # the second call: libraries and allocator warm
            torch.manual_seed(4)
            ls = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bear_net.train(data.repeat(STEPS), 1365, STEPS, 0, "dna", 5, ar_funcs.make_ar_func_cnn, KW, 0.01, "Adam", False, acc_steps=acc,
                           loss_save=ls)
            dt = time.perf_counter() - t0
        res[mode] = (dt / STEPS * 1e6, ls[-1], dict(_train.LAST_RUN))
    if len(MODES) < 2:
        print(acc, res)
        continue
    print(f"acc_steps={acc}: eager {res['eager'][0]:.1f} us/step, captured {res['graph'][0]:.1f} us/step "
          f"({res['eager'][0] / res['graph'][0]:.2f}x; whole train() call incl. upload and capture), last loss {res['eager'][1]:.12g} / {res['graph'][1]:.12g}, "
          f"{res['graph'][2]}")
