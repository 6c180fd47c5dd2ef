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
    cnt = int(w[0])
    ent = w[8:8 + cnt * 8].view(cnt, 8)
    ovf = int((ent[:, 1] == 255).sum())
    nc = torch.tensor([bin(int(v)).count("1") for v in ent[:, 1].tolist() if v != 255])
    print(f"{name}: tokens {n_tok}, work-list {cnt} ({100*cnt/n_tok:.2f}%), overflow {ovf}, "
          f"candidates/hist {torch.bincount(nc, minlength=7).tolist() if len(nc) else []}")
    # gap statistics in fp64
    x64 = x.reshape(-1, bench.D).double(); c64 = cb.double()
    d2 = (c64 * c64).sum(1)[None] - 2 * x64 @ c64.t()
    top2 = d2.topk(2, dim=1, largest=False).values
    gap = (top2[:, 1] - top2[:, 0])
    print("   fp64 gap best->second: median %.3f, 1%% %.4f, min %.2e; |x|*|c|max median %.1f" % (
        gap.median(), gap.kthvalue(max(1, n_tok // 100)).values, gap.min(),
        (x64.norm(dim=1) * c64.norm(dim=1).max()).median()))
