"""GPU diagnostic: class-graph + instance-graph GNN at an ImageNet-like width (embed_dim 1024): split-fp16 MFMA GEMMs
with unfused LayerNorm / pooling against the fp32 library-GEMM route.  python tools/time_gnn_wide.py [E] [K] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import schema_inference.graph as graph
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
M = 1024
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(dev)
g = torch.Generator().manual_seed(1)
nodes = torch.rand(K, n, generator=g).to(dev)
edges = (torch.rand(K, n, n, generator=g) / n).to(dev)
ids = torch.stack([torch.randperm(M, generator=g)[:n] for _ in range(K)]).to(dev)
res = {}
with torch.no_grad():
    for mode in ("1", "0"):
        os.environ["SN_GCN_MFMA"] = mode
        for _ in range(2): out = m.gnn(nodes, edges, ids)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = m.gnn(nodes, edges, ids)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 5 * 1e3, out)
err = (res["1"][1] - res["0"][1]).abs().max().item() / res["0"][1].abs().max().item()
print(f"GNN over {K} graphs of {n} vertices, embed_dim {E}: MFMA route {res['1'][0]:.2f} ms, library route {res['0'][0]:.2f} ms; max rel diff {err:.2e}")
