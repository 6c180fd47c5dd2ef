"""GPU helper for profilers: N launches of sn_assign_words on the bench shape (screen + re-rank), nothing else.
python tools/run_assign.py [n_launches]   (SN_ASSIGN_VARIANT selects the screen form)"""
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
x = tokens[:, 1:, :]
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
torch.cuda.synchronize()
print("ok", int(out.sum()))
