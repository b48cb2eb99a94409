"""Dev helper: bear_plan_create time, tile cut on the device vs on the host (BEAR_PLAN_CUT=host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels
dev = torch.device("cuda", 0)
for n in (100_000_000, 1_000_000_000):
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
    for mode in ("device", "host", "device", "host"):
        if mode == "host": os.environ["BEAR_PLAN_CUT"] = "host"
        else: os.environ.pop("BEAR_PLAN_CUT", None)
        torch.cuda.synchronize(); t0 = time.time()
        p = kernels.Plan(t, 5)
        torch.cuda.synchronize(); dt = time.time() - t0
        print("%.0e contexts, cut on the %s: %.3f s (%d tiles)" % (n, mode, dt, len(p.tiles()[0])), flush=True)
        del p
    del t
