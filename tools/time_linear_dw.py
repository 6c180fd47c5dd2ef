import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schemanet-pytorch_amd"))
import torch
from schema_inference.graph import gnn as g
lin = torch.nn.Linear(256, 256).cuda()
x = torch.randn(101, 1024, 256, device="cuda", requires_grad=True)
dy = torch.randn(101, 1024, 256, device="cuda")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def lib():
    lin.zero_grad(set_to_none=True); x.grad = None
    lin(x).backward(dy)
def per():
    lin.zero_grad(set_to_none=True); x.grad = None
    g._linear(lin, x).backward(dy)
print("library linear fwd+bwd %.3f ms, per-graph dW %.3f ms" % (t(lib), t(per)))
