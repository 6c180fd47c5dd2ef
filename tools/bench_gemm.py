"""GPU micro-benchmark of sn_gcn_gemm: per-launch time for the bench shapes at several batch counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops

dev = torch.device("cuda", 0)
def run(m, n, k, batches, ln, shared_a=False, reps=20):
    A = torch.randn(1 if shared_a else batches, m, k, device=dev)
    Bt = torch.randn(batches, n, k, device=dev)
    a, b = ops.split_planes(A), ops.split_planes(Bt)
    g, be = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    kw = dict(layernorm=(g, be, 1e-5), relu=True) if ln else {}
    for _ in range(3):
        out = ops.gcn_gemm(a, b, batches, want_planes=n, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ops.gcn_gemm(a, b, batches, want_planes=n, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * m * n * k * batches
    blocks = batches * ((m + 127) // 128) * ((n + 255) // 256)
    print(f"m={m} n={n} k={k} batches={batches:4d} ln={int(ln)} blocks={blocks:4d}: {us:8.1f} us  {flops/us/1e6:7.1f} TFLOP/s fp32-equivalent "
          f"({3*flops/us/1e6:7.1f} fp16 MFMA)  {((m if not shared_a else 0)+n)*k*4*batches/us/1e3:7.1f} GB/s operands")

for batches in (1, 16, 64, 100, 128, 256):
    run(512, 256, 512, batches, True)
for batches in (64, 128, 256):
    run(196, 256, 196, batches, True)
for batches in (100, 256):
    run(256, 512 if batches == 100 else 196, 256, batches, False, shared_a=True)

# ---- in-kernel stamps for the atlas-shaped product at 2 workgroups per CU
import ctypes
from cpp_extension import _native as N
lib = N.load()
lib.sn_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]; lib.sn_debug_set_gemm_stamps.restype = None
for (m, n, k, batches) in ((512, 256, 512, 128), (512, 256, 512, 64), (196, 256, 196, 256)):
    A = torch.randn(batches, m, k, device=dev); Bt = torch.randn(batches, n, k, device=dev)
    a, b = ops.split_planes(A), ops.split_planes(Bt)
    g, be = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    blocks = 8 * ((batches + 7) // 8) * ((m + 127) // 128)
    st = torch.zeros(blocks * 4 * 16, dtype=torch.int64, device=dev)
    for _ in range(2):
        ops.gcn_gemm(a, b, batches, want_planes=n, layernorm=(g, be, 1e-5), relu=True)
    lib.sn_debug_set_gemm_stamps(st.data_ptr())
    ops.gcn_gemm(a, b, batches, want_planes=n, layernorm=(g, be, 1e-5), relu=True)
    torch.cuda.synchronize()
    lib.sn_debug_set_gemm_stamps(None)
    s8 = st.view(-1, 16).cpu().double()
    s8 = s8[s8[:, 0] > 0]
    t0 = s8[:, 0].min()
    print(f"m={m} k={k} batches={batches}: span {s8[:,2].max()-t0:.0f} cycles; start skew {s8[:,0].max()-t0:.0f}; "
          f"loop median {(s8[:,1]-s8[:,0]).median():.0f} (wait+barrier {s8[:,3].median():.0f}, dma issue {s8[:,4].median():.0f}); "
          f"epilogue median {(s8[:,2]-s8[:,1]).median():.0f}; stages {k // 16 + (1 if k % 16 else 0)}")
