"""GPU diagnostic: atlas prune + row sums and adjacency operand kernels alone (bench shape)."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench
from cpp_extension import _native as N
dev = torch.device("cuda", 0)
disc, sn, m = bench.make_model(dev)
lib = N.load()
with torch.no_grad():
    fn = lambda: sn.get_atlas(fused_adjacency=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(f"get_atlas(fused_adjacency=True): {(time.perf_counter()-t0)/50*1e6:.1f} us per call")
    lib.sn_profile_enable(20)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    n = lib.sn_profile_count(3); buf = (ctypes.c_float * n)(); lib.sn_profile_elapsed_ms(3, buf, n)
    v = sorted(buf); print(f"prune + row sums kernel: median {v[n//2]*1e3:.1f} us (min {v[0]*1e3:.1f}) = {100*512*512*4/(v[n//2]*1e-3)/1e9:.0f} GB/s read")
    lib.sn_profile_enable(0)
