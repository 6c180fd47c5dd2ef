"""Which Python line launches what in one eager training iteration at config [4]'s real size (torch.profiler with stacks): per kernel
name the launch count and the innermost frames of this package that issued it.   python tools/train_launch_sources.py [compact 0/1]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import schema_inference.graph as graph
from schema_inference import loss as loss_mod, train as train_mod
DEV = "cuda"
compact = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
B, L, M, K, E = 64, 196, 1024, 101, 256
g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
ing = torch.randint(0, M, (B, L), generator=g(1)); ing[:, ::3] = ing[:, :1]
batch = {"ingredients": ing.to(DEV), "attn": torch.randn(B, L, L, generator=g(2)).to(DEV), "attn_cls": torch.randn(B, L, generator=g(3)).to(DEV)}
target = {"label": torch.randint(0, K, (B,), generator=g(4)).to(DEV)}
torch.manual_seed(11)
sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(DEV)
sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
torch.manual_seed(12)
m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV).train()
sn.compact_training = compact


def fwd(b):
    atlas = sn.get_atlas()
    o = {"pred": m.forward_padded(sn.instance_graph_padded(b["ingredients"], b["attn"].clone(), b["attn_cls"].clone()), atlas)}
    o.update(atlas)
    return o


loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
opt = torch.optim.AdamW(list(sn.parameters()) + list(m.parameters()), lr=1e-3, weight_decay=5e-4, fused=True)
for _ in range(3):
    train_mod.train_iter(lambda: fwd(batch), sn, loss_fn, weights, opt, target)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_mod.train_iter(lambda: fwd(batch), sn, loss_fn, weights, opt, target)
    torch.cuda.synchronize()
# CPU-side operator events carry the Python stack: count the operators that launch device work, by the innermost frames of this
# package / its callers (autograd's backward nodes show up under the Function's backward)
by_src = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue                                   # (count the outermost aten op only)
    if e.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::reshape", "aten::detach", "aten::as_strided", "aten::select",
                  "aten::slice", "aten::expand", "aten::t", "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::alias", "aten::permute", "aten::_unsafe_view",
                  "aten::lift_fresh", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::result_type", "aten::unbind", "aten::stride", "aten::size"):
        continue
    frames = [f for f in (e.stack or []) if ("schemanet-pytorch_amd" in f or "tools/" in f) and "train_launch_sources" not in f]
    src = " <- ".join(f.split("schemanet-pytorch_amd/")[-1].strip()[:64] for f in frames[:2]) or "(autograd engine / optimizer)"
    by_src[(e.name, src)] += 1
print("aten operators of one iteration (outermost, views left out), compact_training", compact)
for (name, src), c in sorted(by_src.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{c:4d} x {name:28s} {src}")
