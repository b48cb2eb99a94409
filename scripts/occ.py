import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bear_amd import _lib, kernels
ws = kernels.Workspace()
L = _lib.lib()
print("occupancy blocks/CU: prior", L.bear_debug_occupancy(0), "ref", L.bear_debug_occupancy(1))
p = torch.cuda.get_device_properties(0)
print(p.name, p.multi_processor_count, "shared/block", getattr(p, "shared_memory_per_block", None), getattr(p, "shared_memory_per_block_optin", None), getattr(p, "shared_memory_per_multiprocessor", None))
