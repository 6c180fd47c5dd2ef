"""GPU diagnostic: the class branch (atlas prune / row sums -> adjacency planes -> GNN over the K class graphs) run on all
classes at once and in groups of classes whose atlas slab + operand planes fit the 256 MB Infinity Cache.
python tools/time_class_chunks.py K n E [groups ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import schema_inference.graph as graph
from cpp_extension import ops
dev = torch.device("cuda", 0)
K, n, E = (int(v) for v in sys.argv[1:4])
groups = [int(v) for v in sys.argv[4:]] or [1, 2, 4, 8]
M = n
torch.manual_seed(4)
sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(dev)
sn.register_class_vertices(torch.arange(M, device=dev).repeat(K, 1))
torch.manual_seed(5)
m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(dev)
vw, ew, ids = sn.vertex_weights.tensor.detach(), sn.edge_weights.tensor.detach(), sn.class_ingredients.tensor
prepared = m.gnn.prepare()


def branch(g):
    outs = []
    step = (K + g - 1) // g
    for k0 in range(0, K, step):
        cv, adj = ops.atlas_adjacency_planes(vw[k0:k0 + step], ew[k0:k0 + step], 0.001, False)
        outs.append(m.gnn(nodes=cv, edges=None, ingredients=ids[k0:k0 + step], adjacency=adj, prepared=prepared))
    return torch.cat(outs) if len(outs) > 1 else outs[0]


def t(fn, reps=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


with torch.no_grad():
    ref = branch(1)
    for g in groups:
        out = branch(g)
        print("K=%d n=%d E=%d atlas %.0f MB: %d group(s) %.0f us  (max |diff| vs one group %.1e)" % (K, n, E, K * n * n * 4 / 1e6, g, t(lambda: branch(g)), float((out - ref).abs().max())))
