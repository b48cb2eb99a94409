"""Dev: configs[1]'s / configs[3]'s step (bear_ref, stop prior) against the grid of dm_ref_items_kernel (a library built with the
BEAR_DEV_REF_GRID override of launch_ref_plan, passed as BEAR_AMD_LIB): blocks per CU = the argument."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import baseline_configs
dev = torch.device("cuda", 0)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for g in ("8", "6", "4", "3", "2"):
    os.environ["BEAR_DEV_REF_GRID"] = g
    out = baseline_configs.measure_configs(dev, only=which)
    print(which, g, [round(v["us_per_step"], 2) for k, v in out.items() if isinstance(v, dict)], flush=True)
