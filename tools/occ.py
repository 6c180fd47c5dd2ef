import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "schemanet-pytorch_amd")]
import torch
torch.cuda.init()
from cpp_extension import _native as N
lib = N.load()
lib.sn_debug_screen_occupancy.argtypes = [ctypes.c_int]; lib.sn_debug_screen_occupancy.restype = ctypes.c_int
for lds in (16384, 32768, 51200, 65536, 70000, 76800, 81920, 90000):
    print(lds, lib.sn_debug_screen_occupancy(lds))
p = torch.cuda.get_device_properties(0)
print(p.name, p.multi_processor_count, getattr(p, "shared_memory_per_block", None), getattr(p, "shared_memory_per_multiprocessor", None), getattr(p, "max_threads_per_multi_processor", None))
