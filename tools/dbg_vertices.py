import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "schemanet-pytorch_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from cpp_extension import ops
from oracle import cabi
g = dict(np.load(os.path.join(ROOT, "tests/golden/ext_small.npz")))
dev = "cuda"
ing = torch.from_numpy(g["ing"]).to(dev); acls = torch.from_numpy(g["attn_cls"]).to(dev); w = torch.from_numpy(g["w_v"]).to(dev)
out = ops.instance_graph(ing, None, acls, w_v=w, n_pad=36, pad_id=-1, attn_cls_is_logits=False, mean=True, want_attr2=True, want_weighted=True)
ids, a2, wts, num_v = cabi.instance_v(g["ing"], g["attn_cls"], g["w_v"], mean=True)
n = out["n"].tolist(); print("n", n, num_v.tolist())
o = 0
for b in range(len(n)):
    got = out["v2"][b, :n[b]].cpu().numpy(); want = a2[o:o + n[b]]; o += n[b]
    print(b, "count col max err", np.abs(got[:, 0] - want[:, 0]).max(), "attn col max err", np.abs(got[:, 1] - want[:, 1]).max(), "got[:3]", got[:3].tolist(), "want[:3]", want[:3].tolist())
