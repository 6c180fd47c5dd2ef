"""GPU diagnostic: the one-round K-outer screen (variant 4) against the exact kernel at the bench shape - where do they differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import cpp_extension
from cpp_extension import ops
lib = cpp_extension.load()
B, L, D, M = 256, 196, 384, 512
g = torch.Generator().manual_seed(0)
tokens = torch.randn(B, L + 1, D, generator=g).cuda()
cb = torch.randn(M, D, generator=torch.Generator().manual_seed(1)).cuda()
cbt, packed = ops.PackedCodebook().get(cb)
exact = ops.assign_words(tokens[:, 1:, :], cbt, packed, mode=1)
lib.sn_assign_set_variant(4)
fast = ops.assign_words(tokens[:, 1:, :], cbt, packed, mode=0)
bad = (fast != exact)
print("mismatches", int(bad.sum()), "of", bad.numel())
print("by position l (nonzero):", {int(l): int(c) for l, c in enumerate(bad.sum(0).tolist()) if c})
idx = bad.nonzero()[:10]
for b, l in idx.tolist():
    x = tokens[b, 1 + l].double(); d = ((cb.double() - x) ** 2).sum(1)
    print(b, l, "fast", int(fast[b, l]), "exact", int(exact[b, l]), "d_fast - d_min", float(d[fast[b, l]] - d.min()), "rank", int((d < d[fast[b, l]]).sum()))
