"""GPU diagnostic: work-list statistics of the S1 screen (how many tokens need the fp64 re-rank,
how many overflow to a full scan) and the screen's actual error vs its rigorous bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench

dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
for name, tok in (("randn tokens", tokens), ("k-means-like", None)):
    if tok is None:
        near = torch.randint(0, bench.M, (bench.B, bench.L), device=dev)
        tok = tokens.clone(); tok[:, 1:, :] = codebook[near] + 0.3 * tokens[:, 1:, :]
    cb, packed = ops.PackedCodebook().get(codebook)
    lib = N.load()
    x = tok[:, 1:, :]
    n_tok = x.shape[0] * x.shape[1]
    ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
    out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)))
    torch.cuda.synchronize()
    w = ws.view(torch.int32)
    flags = w[8:8 + n_tok]
    ovf = int((flags < 0).sum())
    fl = flags[flags > 0]
    cnt = int(fl.numel())
    nc = torch.tensor([bin(int(v)).count("1") for v in fl.tolist()])
    print(f"{name}: tokens {n_tok}, flagged {cnt} ({100*cnt/n_tok:.2f}%), overflow {ovf}, "
          f"candidates/hist {torch.bincount(nc, minlength=7).tolist() if len(nc) else []}")
    # gap statistics in fp64
    x64 = x.reshape(-1, bench.D).double(); c64 = cb.double()
    d2 = (c64 * c64).sum(1)[None] - 2 * x64 @ c64.t()
    top2 = d2.topk(2, dim=1, largest=False).values
    gap = (top2[:, 1] - top2[:, 0])
    print("   fp64 gap best->second: median %.3f, 1%% %.4f, min %.2e; |x|*|c|max median %.1f" % (
        gap.median(), gap.kthvalue(max(1, n_tok // 100)).values, gap.min(),
        (x64.norm(dim=1) * c64.norm(dim=1).max()).median()))

# ---- in-kernel stamps of the screen kernel (shader cycles)
import ctypes
lib.sn_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.sn_debug_set_stamps.restype = None
n_waves = 4 * max((n_tok + 127) // 128, 2 * torch.cuda.get_device_properties(dev).multi_processor_count)
st = torch.zeros(n_waves * 16, dtype=torch.int64, device=dev)
lib.sn_debug_set_stamps(st.data_ptr())
x = tokens[:, 1:, :]
for _ in range(3):
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)))
torch.cuda.synchronize()
lib.sn_debug_set_stamps(None)
s8 = st.view(n_waves, 16).cpu().double()
s8 = s8[(s8[:, 0] > 0) & (s8[:, 1] > 0)]           # waves that ran and had tokens
t0 = s8[:, 0].min()
print("screen kernel stamps (cycles, 100MHz-ref memtime?):")
print("  kernel span (first start -> last end): %.0f" % (s8[:, 3].max() - t0))
print("  start skew  (last wave start - first): %.0f" % (s8[:, 0].max() - t0))
print("  token load+convert: median %.0f  max %.0f" % ((s8[:, 1] - s8[:, 0]).median(), (s8[:, 1] - s8[:, 0]).max()))
print("  main loop:          median %.0f  max %.0f" % ((s8[:, 2] - s8[:, 1]).median(), (s8[:, 2] - s8[:, 1]).max()))
print("     of which dma-wait + barrier: median %.0f max %.0f; issuing the next tile's DMA: median %.0f max %.0f" % (
    s8[:, 4].median(), s8[:, 4].max(), s8[:, 5].median(), s8[:, 5].max()))
print("  tail:               median %.0f" % ((s8[:, 3] - s8[:, 2]).median()))
print("  per-wave total:     median %.0f  max %.0f" % ((s8[:, 3] - s8[:, 0]).median(), (s8[:, 3] - s8[:, 0]).max()))

# ---- kernel times (HIP events) for both token sets, and overflow counts
import ctypes as C
for name, tok in (("randn", tokens), ("k-means-like", tok)):
    x = tok[:, 1:, :]
    lib.sn_profile_enable(20)
    for _ in range(10):
        N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                    N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)))
    torch.cuda.synchronize()
    res = {}
    for kid, kname in ((0, "screen"), (1, "rerank")):
        n = lib.sn_profile_count(kid); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(kid, buf, n)
        res[kname] = sorted(buf)[n // 2] * 1e3
    w = ws.view(torch.int32)
    print(f"{name}: screen {res['screen']:.1f} us, rerank {res['rerank']:.1f} us, flagged {int((w[8:8 + n_tok] > 0).sum())}, overflow tokens {int(w[1])}")
    lib.sn_profile_enable(0)
