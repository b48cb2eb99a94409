"""Developer: which part of eval_plan_kernel costs what (needs the -DEVP_DEBUG_SWITCHES build, BEAR_AMD_LIB=build_variants/libbear_evpdbg.so).
Flags: 2 = no tie resolution, 4 = no cell units, 8 = no total units (results are then meaningless)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bear_amd import kernels

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda")
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "test"))
f = kernels.synth_prior(20211012, 0, n, dev)
plan = kernels.EvalPlan(t["test"], t["train"])


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for flags in (0, 2, 4, 8, 6, 10, 12, 14):
    os.environ["BEAR_EVP_DEBUG"] = str(flags)
    a = timed(lambda: kernels.evaluate_planned(plan, f, [1.0], [0.1, 1.0, 10.0]))
    b = timed(lambda: kernels.evaluate_planned(plan, f, [1.0], None))
    print(f"flags {flags:2d}: 1h+AR+3van {a:7.3f} ms   1h+AR {b:7.3f} ms")
