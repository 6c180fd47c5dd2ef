"""The autograd graph of one training forward at config [4]'s real size: node types counted, and for the small library nodes (select, add,
mul, sum, where, ...) the shapes they work on - which of the ~145 library launches of an iteration come from where.
    python tools/train_graph_nodes.py [compact 0/1]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
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
loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
sn.normalize()
atlas = sn.get_atlas()
out = {"pred": m.forward_padded(sn.instance_graph_padded(batch["ingredients"], batch["attn"].clone(), batch["attn_cls"].clone()), atlas)}
out.update(atlas)
ld = loss_fn(out, target)
loss = train_mod.weighted_total(ld, weights)
seen, order = set(), []
stack = [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n); order.append(n)
    for nxt, _ in n.next_functions:
        stack.append(nxt)
cnt = collections.Counter(type(n).__name__ for n in order)
print("autograd nodes:", len(order))
for k_, v in cnt.most_common(40):
    print(f"{v:4d} x {k_}")
print("--- shapes of the small library nodes")
for n in order:
    nm = type(n).__name__
    if nm.startswith(("Select", "Slice", "Index", "Unsqueeze", "Squeeze", "Expand", "Sum", "Mul", "Add", "Where", "Div", "Neg", "Gather", "Scatter", "Max", "Clone", "Copy", "Cat", "Pad", "Constant")):
        shp = None
        for attr in ("_saved_self_sym_sizes", "_saved_self", "_saved_other", "_saved_mask"):
            if hasattr(n, attr):
                try:
                    v = getattr(n, attr)
                    shp = tuple(v.shape) if torch.is_tensor(v) else tuple(v)
                    break
                except Exception:
                    pass
        nxt = [type(f).__name__ for f, _ in n.next_functions if f is not None]
        print(f"{nm:28s} {str(shp):28s} <- {', '.join(nxt)[:120]}")
