"""GPU diagnostic: the fused LayerNorm passes of the wide GCN route at config [3]'s class-side shape (1000 x 500 x 1024)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops
dev = torch.device("cuda", 0)
G, n, E = 1000, 500, 1024
x = torch.randn(G, n, E, device=dev)
gamma = torch.rand(E, device=dev) + 0.5; beta = torch.randn(E, device=dev) * 0.1
nodes = torch.rand(G, n, device=dev)
scale = ops.pow2_scale(16.0 * gamma.abs().max() + beta.abs().max())


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for relu in [int(v) for v in (sys.argv[1:] or ["1"])]:
    print("relu flag", relu, "ln+split %.0f us" % t(lambda: ops.layernorm_split_planes(x, gamma, beta, 1e-5, relu=relu, scale=scale)))
print("ln+pool %.0f us" % t(lambda: ops.layernorm_weighted_pool(x, gamma, beta, 1e-5, nodes, relu=True)))
y = x.clone()
print("ln in place %.0f us" % t(lambda: ops.mask_layernorm_act_(y, gamma, beta, 1e-5, relu=True)))
print("split %.0f us" % t(lambda: ops.split_planes(y, scale=scale)))
print("copy %.0f us" % t(lambda: y.copy_(x)))
