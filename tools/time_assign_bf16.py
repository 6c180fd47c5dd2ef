"""GPU diagnostic: S1 on bfloat16 tokens consumed in place against the fp32 tokens of the bench (HIP events inside the library)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
lib = N.load()
cb, packed = ops.PackedCodebook().get(codebook)
for name, tok in (("fp32", tokens), ("bf16", tokens.to(torch.bfloat16))):
    x = tok[:, 1:, :]
    for _ in range(100): out = ops.assign_words(x, cb, packed)
    torch.cuda.synchronize()
    lib.sn_profile_enable(100)
    for _ in range(100): out = ops.assign_words(x, cb, packed)
    torch.cuda.synchronize()
    res = {}
    for kid, kname in ((0, "screen"), (1, "rerank")):
        n = lib.sn_profile_count(kid); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(kid, buf, n); res[kname] = sorted(buf)[n // 2] * 1e3
    lib.sn_profile_enable(0)
    exact = ops.assign_words(x, cb, packed, mode=1)
    print(f"{name} tokens: screen {res['screen']:.1f} us, re-rank {res['rerank']:.1f} us, mismatches vs exact {int((out != exact).sum())}", flush=True)
