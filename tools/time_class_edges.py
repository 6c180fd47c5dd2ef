import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schemanet-pytorch_amd"))
import torch
import schema_inference.graph as graph
dev = "cuda"
M, K = 1024, 101
torch.manual_seed(11)
sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(dev)
sn.register_class_vertices(torch.arange(M, device=dev).repeat(K, 1))
sn.train()
gy = torch.randn(K, M, M, device=dev)
def fb():
    sn.edge_weights.tensor.grad = None
    ce = sn.get_class_edges()
    ce.backward(gy)
for _ in range(3): fb()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fb()
torch.cuda.synchronize(); print("get_class_edges fwd+bwd %.2f ms" % ((time.perf_counter() - t0) * 100))
def f():
    with torch.no_grad():
        return sn.get_atlas()
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize(); print("no-grad get_atlas (HIP) %.2f ms" % ((time.perf_counter() - t0) * 100))
