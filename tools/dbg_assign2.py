"""GPU debug: mismatches of the screen2 path vs the exact kernel, with flag words."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from cpp_extension import ops, _native as N
import datagen
dev = torch.device("cuda", 0)
lib = N.load()
for (D, M, B) in ((192, 128, 8), (384, 512, 12), (384, 100, 4)):
    L = 196
    cb = torch.from_numpy(datagen.bellish((M, D), 7 + D, 1.0)).to(dev)
    x = torch.from_numpy(datagen.bellish((B, L, D), 9 + M, 1.0)).to(dev)
    cbt, packed = ops.PackedCodebook().get(cb)
    n_tok = B * L
    ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
    out = torch.full((B, L), -1, dtype=torch.int64, device=dev)
    ex = torch.empty_like(out)
    for mode, o in ((0, out), (1, ex)):
        N.check(lib.sn_assign_words(N.ptr(x), B, L, x.stride(0), x.stride(1), N.ptr(cbt), N.ptr(packed), M, D,
                                    N.ptr(o), o.stride(0), o.stride(1), N.ptr(ws), ws.numel(), mode, N.stream_ptr(dev)), "assign")
        torch.cuda.synchronize()
        if mode == 0:
            fl = ws[32:32 + 4 * n_tok].view(torch.int32).clone()
    bad = (out != ex).reshape(-1).nonzero().reshape(-1)
    print(f"D={D} M={M}: {bad.numel()} / {n_tok} mismatches; flagged {(fl > 0).sum().item()} overflow {(fl < 0).sum().item()} unwritten {(out < 0).sum().item()}")
    o1, e1 = out.reshape(-1), ex.reshape(-1)
    for t in bad[:24].tolist():
        print(f"   token {t} (set {t // 16}, tau {t % 16}): got {o1[t].item()} want {e1[t].item()} flag {fl[t].item():#x}")
