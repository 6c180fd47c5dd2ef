"""GPU micro-benchmark of the plane producers (adjacency, gather) with and without the device extent."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops
dev = torch.device("cuda", 0)
def t(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
edges = torch.rand(256, 196, 196, device=dev)
ext = torch.tensor([125], dtype=torch.int32, device=dev)
print("instance adjacency planes: full %.1f us, extent 125: %.1f us" % (t(lambda: ops.gcn_adjacency_planes(edges)), t(lambda: ops.gcn_adjacency_planes(edges, extent=ext))))
table = torch.randn(513, 256, device=dev); ids = torch.randint(0, 513, (256, 196), device=dev)
print("instance gather planes:    full %.1f us, extent 125: %.1f us" % (t(lambda: ops.gcn_gather_planes(table, ids)), t(lambda: ops.gcn_gather_planes(table, ids, extent=ext))))
vw = torch.rand(100, 512, device=dev); ew = torch.randn(100, 512, 512, device=dev)
print("atlas prune + rowsum + fused adjacency: %.1f us; plain adjacency of [100,512,512]: %.1f us" % (
    t(lambda: ops.atlas_adjacency_planes(vw, ew, 0.001, False)), t(lambda: ops.gcn_adjacency_planes(ew))))
ew2 = torch.randn(101, 1024, 1024, device=dev)
print("plain adjacency of [101,1024,1024]: %.1f us" % t(lambda: ops.gcn_adjacency_planes(ew2)))
# the compacted producer on a 70 % pruned atlas (bench shape)
g = torch.Generator().manual_seed(44)
vwp = torch.rand(100, 512, generator=g).to(dev)
low = (torch.rand(100, 512, generator=g) < 0.7).to(dev)
vwp = torch.where(low, vwp * 1e-4, vwp)
ewp = torch.rand(100, 512, 512, generator=g).to(dev)
ops.atlas_adjacency_planes_compact(vwp, ewp, 0.001, False)
print("compacted atlas route (70 %% pruned): prune + rowsum + keep_perm + producer %.1f us; with the rows known zero %.1f us" % (
    t(lambda: ops.atlas_adjacency_planes_compact(vwp, ewp, 0.001, False)), t(lambda: ops.atlas_adjacency_planes_compact(vwp, ewp, 0.001, False, pruned_rows_are_zero=True))))
