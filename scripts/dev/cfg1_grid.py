"""Dev: configs[1]'s step (bear_ref, stop prior, 1e7 contexts) against the grid of dm_ref_items_kernel (a library built with the
BEAR_DEV_REF_GRID override, passed as BEAR_AMD_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import baseline_configs
dev = torch.device("cuda", 0)
for g in ("8", "4", "3", "2", "1"):
    os.environ["BEAR_DEV_REF_GRID"] = g
    out = baseline_configs.measure_configs(dev, only=1)
    print(g, [round(v["us_per_step"], 2) for k, v in out.items() if isinstance(v, dict)], flush=True)
