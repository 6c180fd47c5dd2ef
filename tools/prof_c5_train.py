"""GPU diagnostic: kernel breakdown of training iterations at config [4]'s real size (B = 64, M = 1024, K = 101, n_max = 1024,
E = 256; the model of tests/test_gpu_api.py::test_c5_real_size_training_iterations) - run under `rocprofv3 --kernel-trace
--stats` (from /tmp), or alone for wall times.  python tools/prof_c5_train.py [iterations] [mfma 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ROUTE = sys.argv[4] if len(sys.argv) > 4 else "padded"
os.environ["SN_GCN_MFMA"] = sys.argv[2] if len(sys.argv) > 2 else "1"
import schema_inference.graph as graph
from schema_inference import loss as loss_mod, train as train_mod
DEV = "cuda"
B, L, D, M, K, E = 64, 196, 384, 1024, 101, 256
g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
ing = torch.randint(0, M, (B, L), generator=g(1)); ing[:, ::3] = ing[:, :1]
attn = torch.randn(B, L, L, generator=g(2)); acls = torch.randn(B, L, generator=g(3)); label = torch.randint(0, K, (B,), generator=g(4))
torch.manual_seed(11)
sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(DEV)
sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
torch.manual_seed(12)
m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)


class Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.schema_net, self.matcher = sn, m

    def forward(self, batch):
        atlas = self.schema_net.get_atlas()
        if ROUTE == "lists":          # the reference's python lists (one host synchronisation for the sizes)
            inst = self.schema_net(batch["ingredients"], batch["attn"].clone(), batch["attn_cls"].clone())
            out = {"pred": self.matcher(inst, atlas)}
        else:                         # what SchemaNetPredictor.forward takes: the padded batch, no synchronisation
            g_ = self.schema_net.instance_graph_padded(batch["ingredients"], batch["attn"].clone(), batch["attn_cls"].clone())
            out = {"pred": self.matcher.forward_padded(g_, atlas)}
        out.update(atlas)
        return out


model = Model().train()
loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
FUSED = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=5e-4, fused=FUSED, capturable=ROUTE == "graphed")
batch = {"ingredients": ing.to(DEV), "attn": attn.to(DEV), "attn_cls": acls.to(DEV)}
target = {"label": label.to(DEV)}
weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
times, losses = [], []
step = train_mod.GraphedTrainIter(model, model.schema_net, loss_fn, weights, opt, batch, target) if ROUTE == "graphed" else None
for it in range(n_it):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if step is not None:
        total, _ = step(batch, target)
    else:
        total, _ = train_mod.train_iter(lambda: model(batch), model.schema_net, loss_fn, weights, opt, target)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    losses.append(round(float(total), 5))
print("losses", losses, "memset nodes replaced / left", (step.memsets_replaced, step.memsets_left) if step is not None else None)
print("route", ROUTE, "mfma", os.environ["SN_GCN_MFMA"], "fused AdamW", FUSED, "ms per iteration", [round(1e3 * t, 2) for t in times], "loss", float(total))
