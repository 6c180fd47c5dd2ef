"""GPU diagnostic: S1 kernel times (HIP events inside the library) and re-rank statistics on the
bench shape, for the screen variant selected by SN_ASSIGN_VARIANT (2 = register-stationary, 0/1 =
token-stationary).  python tools/time_assign.py [n_launches]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench

dev = torch.device("cuda", 0)
variant = int(os.environ.get("SN_ASSIGN_VARIANT", "2"))
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tokens, codebook, attn = bench.make_inputs(0, dev)
lib = N.load()
cb, packed = ops.PackedCodebook().get(codebook)
near = torch.randint(0, bench.M, (bench.B, bench.L), device=dev)
tok_km = tokens.clone(); tok_km[:, 1:, :] = codebook[near] + 0.3 * tokens[:, 1:, :]
for name, tok in (("randn", tokens), ("k-means-like", tok_km)):
    x = tok[:, 1:, :]
    n_tok = x.shape[0] * x.shape[1]
    ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
    out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)

    def run():
        N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                    N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.sn_profile_enable(n_launch)
    for _ in range(n_launch):
        run()
    torch.cuda.synchronize()
    res = {}
    for kid, kname in ((0, "screen"), (1, "rerank")):
        n = lib.sn_profile_count(kid); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(kid, buf, n)
        v = sorted(buf)
        res[kname] = (v[n // 2] * 1e3, v[0] * 1e3)
    lib.sn_profile_enable(0)
    fl = ws[32:32 + 4 * n_tok].view(torch.int32)
    over = int((fl < 0).sum()); flagged = int((fl > 0).sum())
    exact = torch.empty_like(out)
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(exact), exact.stride(0), exact.stride(1), N.ptr(ws), ws.numel(), 1, N.stream_ptr(dev)), "assign exact")
    torch.cuda.synchronize()
    bad = int((exact != out).sum())
    alg = n_tok * (bench.D * 4 + 8)
    print(f"variant {variant} {name}: screen {res['screen'][0]:.1f} us (min {res['screen'][1]:.1f}) = {alg / res['screen'][0] / 1e3:.0f} GB/s "
          f"= {alg / res['screen'][0] / 1e3 / 80:.1f}% of 8 TB/s; rerank {res['rerank'][0]:.1f} us; flagged {flagged} ({100 * flagged / n_tok:.2f}%), "
          f"overflow {over}; mismatches vs exact kernel {bad}", flush=True)

if variant == 2:      # in-kernel stamps of the register-stationary screen (shader clock ticks)
    lib.sn_debug_set_stamps.argtypes = [C.c_void_p]; lib.sn_debug_set_stamps.restype = None
    n_waves = 4 * 256
    st = torch.zeros(n_waves * 16, dtype=torch.int64, device=dev)
    lib.sn_debug_set_stamps(st.data_ptr())
    x = tokens[:, 1:, :]
    for _ in range(3):
        N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                    N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
    torch.cuda.synchronize()
    lib.sn_debug_set_stamps(None)
    s8 = st.view(n_waves, 16).cpu().double()
    s8 = s8[s8[:, 0] > 0]
    t0 = s8[:, 0].min()
    q = lambda v: "median %.0f max %.0f" % (v.median(), v.max())
    print("stamps (s_memtime ticks): kernel span %.0f; start skew %.0f" % (s8[:, 2].max() - t0, s8[:, 0].max() - t0))
    print("  prologue (codebook -> registers, set 0 converted): " + q(s8[:, 1] - s8[:, 0]))
    print("  loop: " + q(s8[:, 2] - s8[:, 1]) + "  sets per workgroup: " + q(s8[:, 7]))
    print("    MFMA stream: " + q(s8[:, 4]) + "; wait for DMA/LDS: " + q(s8[:, 5]) + "; barrier: " + q(s8[:, 6]))

if variant == 3:      # in-kernel stamps of the K-outer screen (shader clock ticks): 0 start, 1 prologue done, 2 loop done, 3 records written
    lib.sn_debug_set_stamps.argtypes = [C.c_void_p]; lib.sn_debug_set_stamps.restype = None
    n_waves = 16 * 256
    st = torch.zeros(n_waves * 16, dtype=torch.int64, device=dev)
    lib.sn_debug_set_stamps(st.data_ptr())
    x = tokens[:, 1:, :]
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
    torch.cuda.synchronize()
    lib.sn_debug_set_stamps(None)
    s8 = st.view(n_waves, 16).cpu().double()
    s8 = s8[s8[:, 0] > 0]
    t0 = s8[:, 0].min()
    q = lambda v: "median %.0f max %.0f" % (v.median(), v.max())
    print("stamps (s_memtime ticks): %d waves; kernel span %.0f; start: %s" % (s8.shape[0], s8[:, 3].max() - t0, q(s8[:, 0] - t0)))
    print("  prologue: " + q(s8[:, 1] - s8[:, 0]) + "   loop: " + q(s8[:, 2] - s8[:, 1]) + "   keys + records: " + q(s8[:, 3] - s8[:, 2]))
    if s8[:, 12].max() > 0:
        print("  round 0: start -> hcs staged %s | its barrier %s | acc init + wait first chunk %s | barrier %s | read + convert %s | loop %s | keys + records %s" % (
            q(s8[:, 6] - s8[:, 0]), q(s8[:, 7] - s8[:, 6]), q(s8[:, 8] - s8[:, 7]), q(s8[:, 11] - s8[:, 8]), q(s8[:, 12] - s8[:, 11]), q(s8[:, 13] - s8[:, 12]), q(s8[:, 14] - s8[:, 13])))
        print("  round 1: prologue %s" % q(s8[:, 1] - s8[:, 14]))
    print("  (stamps 1-3 are those of the LAST round of a workgroup; whole workgroup: " + q(s8[:, 3] - s8[:, 0]) + ")")
