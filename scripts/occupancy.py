"""Resident workgroups per CU of the traversal kernels (needs a -DRT_TRACE_STATS=1 variant library)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from ray_tracing_amd import api
out = (C.c_int * 4)()
rc = api.lib.rt_debug_occupancy(out)
print(os.path.basename(os.environ.get("RT_AMD_LIB", "default")), "rc", rc, "blocks/CU: extend", out[0], "shadow", out[1], "extend_lean", out[2], "shade", out[3])
