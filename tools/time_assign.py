"""GPU diagnostic: S1 kernel times (HIP events inside the library) and re-rank statistics on the
bench shape, for the screen form selected by SN_ASSIGN_VARIANT (0 = token-stationary, the default; 5 = K-outer one-round;
tools/time_s1.py times both in one process).  python tools/time_assign.py [n_launches]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench

dev = torch.device("cuda", 0)
variant = int(os.environ.get("SN_ASSIGN_VARIANT", "0"))
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
# (the in-kernel stamps of the K-outer one-round form: tools/diag_s5.py; of the default form: tools/diag_assign.py)
