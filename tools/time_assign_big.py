"""GPU diagnostic: S1 at the reference's ImageNet codebook size (M = 8000 words, D = 384, 256 images): MFMA screen +
re-rank (10-bit word codes) against the exact kernel.  python tools/time_assign_big.py [M]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
tokens = torch.randn(256, 197, 384, generator=g).to(dev)
pool = torch.randn(4 * M, 384, generator=torch.Generator().manual_seed(1))
cb = (pool[torch.randperm(4 * M, generator=torch.Generator().manual_seed(2))[:M]]).to(dev)
cbt, packed = ops.PackedCodebook().get(cb)
x = tokens[:, 1:, :]
def t(mode, n):
    for _ in range(2): out = ops.assign_words(x, cbt, packed, mode=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = ops.assign_words(x, cbt, packed, mode=mode)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
ms_fast, fast = t(0, 20)
ms_exact, exact = t(1, 2)
print(f"M={M}: screen + re-rank {ms_fast:.3f} ms, exact kernel {ms_exact:.2f} ms, mismatches {int((fast != exact).sum())}")
# work-list statistics of the last fast launch
import ctypes as C
from cpp_extension import _native as N
lib = N.load()
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
lib.sn_profile_enable(4)
for _ in range(4):
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cbt), N.ptr(packed), M, 384,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
torch.cuda.synchronize()
res = {}
for kid, kname in ((0, "screen"), (1, "rerank")):
    n = lib.sn_profile_count(kid); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(kid, buf, n); res[kname] = sorted(buf)[n // 2] * 1e3
lib.sn_profile_enable(0)
fl = ws[32:32 + 4 * n_tok].view(torch.int32)
print(f"   screen {res['screen']:.0f} us, re-rank {res['rerank']:.0f} us; flagged {int((fl > 0).sum())} ({100.0 * int((fl > 0).sum()) / n_tok:.1f} %), overflow {int((fl < 0).sum())}")
