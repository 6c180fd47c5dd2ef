import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
lib = N.load()
cb, packed = ops.PackedCodebook().get(codebook)
near = torch.randint(0, bench.M, (bench.B, bench.L), device=dev)
tok = tokens.clone(); tok[:, 1:, :] = codebook[near] + 0.3 * tokens[:, 1:, :]
x = tok[:, 1:, :]
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
for rep in range(4):
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
    torch.cuda.synchronize()
    print("rep", rep, "flagged", int((ws[32:32 + 4 * n_tok].view(torch.int32) > 0).sum()))
fl = ws[32:32 + 4 * n_tok].view(torch.int32).cpu()
idx = (fl > 0).nonzero().reshape(-1)
print("flagged", idx.numel(), "hist of token%32:", torch.bincount(idx % 32, minlength=32).tolist())
print("hist of set%8:", torch.bincount((idx // 32) % 8, minlength=8).tolist())
print("first flags:", [hex(int(v)) for v in fl[idx[:12]].tolist()], idx[:12].tolist())
codes = ws[32 + 4 * n_tok: 32 + 36 * n_tok].view(torch.int32).view(n_tok, 8).cpu()
for t in idx[:4].tolist():
    print(t, "out", int(out.reshape(-1)[t]), "near", int(near.reshape(-1)[t]), [hex(int(v)) for v in codes[t].tolist()])
