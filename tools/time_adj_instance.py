import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
edges = torch.rand(256, 196, 196, generator=g).to(dev)
nv = torch.randint(93, 126, (256,), generator=g, dtype=torch.int32).to(dev)
ext = nv.max().reshape(1).to(torch.int32)
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
a = ops.gcn_adjacency_planes(edges, extent=ext, n_valid=nv.contiguous(), per_graph=True)
print("SN_ADJ_GRAPH_MAJOR", os.environ.get("SN_ADJ_GRAPH_MAJOR", "1"), "instance adjacency planes, per-graph extents: %.1f us (launch to launch); checksum %.6f %.6f" % (
    t(lambda: ops.gcn_adjacency_planes(edges, extent=ext, n_valid=nv.contiguous(), per_graph=True)), a.hi.float().abs().sum().item(), a.lo.float().abs().sum().item()))
