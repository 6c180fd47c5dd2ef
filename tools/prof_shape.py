"""GPU diagnostic: EAGER steps at the shape of config [3] (c4) or [4] (c5), for `rocprofv3 --kernel-trace --stats` (run from
/tmp): every launch a record, the class branch in line behind S1 so that S1 runs alone on the GPU.
    python3 tools/prof_shape.py c4 [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import bench
import discretization
import schema_inference.graph as graph

SHAPES = {"c4": (256, 768, 1024, 1000, 500, 1024, torch.bfloat16), "c5": (64, 384, 1024, 101, 1024, 256, torch.float32),
          "c2": (256, 384, 512, 100, 512, 256, torch.float32)}
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
defer = len(sys.argv) > 3 and sys.argv[3] == "defer"     # the one-at-a-time form of the step: S1's fp64 finish inside the instance-graph kernel
Bc, Dc, Mc, Kc, n_max, Ec, dt = SHAPES[name]
dev = torch.device("cuda", 0)
g = lambda s_: torch.Generator().manual_seed(s_)  # noqa: E731
pool = torch.randn(4 * Mc, Dc, generator=g(1))
codebook = (pool[torch.randperm(4 * Mc, generator=g(11))[:Mc]] + 0.05 * torch.randn(Mc, Dc, generator=g(2))).to(dev)
torch.manual_seed(4)
sn = graph.SchemaNet(num_vertices=Mc, num_classes=Kc, dist_pow=2, feat_h=14, feat_w=14, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0,
                     remove_self_loop=False, prune_node_threshold=0.001, class_max_vertices=n_max)
sn.register_class_vertices(torch.stack([torch.randperm(Mc, generator=g(20 + k))[:n_max] for k in range(Kc)]))
torch.manual_seed(5)
m = graph.Matcher("inner_product", Mc, dict(embed_dim=Ec, num_layers=2, identity_proj=False, activation="relu"))
disc = discretization.Discretization(Mc, Dc)
disc, sn, m = disc.to(dev), sn.to(dev), m.to(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    tk = torch.randn(Bc, bench.L + 1, Dc, generator=g(200)).to(dev, dt)
    at = torch.randn(Bc, bench.L + 1, bench.L + 1, generator=g(203)).to(dev)
    run = lambda: bench.step(disc, sn, m, tk, at, class_branch_first=False, side_stream=False, defer=defer)      # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        last = run()
    torch.cuda.synchronize()
    dt_ms = 1e3 * (time.perf_counter() - t0) / n
print(f"{name}{' (deferred S1 finish)' if defer else ''}: {n} eager steps in line, {dt_ms:.3f} ms per step, {Bc / dt_ms * 1e3:.0f} img/s", tuple(last.shape))
