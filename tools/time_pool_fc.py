import torch, sys
sys.path.insert(0, "schemanet-pytorch_amd")
from cpp_extension import ops
dev = torch.device("cuda", 0)
for G, parts in [(256, 1), (100, 4)]:
    pooled = torch.randn(G, parts, 256, device=dev); W = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev)
    Wt = W.t().contiguous(); div = torch.tensor([125], dtype=torch.int32, device=dev)
    for name, kw in [("rows", {}), ("transposed", {"weight_t": Wt})]:
        for _ in range(5): ops.pool_fc(pooled, div, W, b, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): ops.pool_fc(pooled, div, W, b, **kw)
        e1.record(); torch.cuda.synchronize()
        print(G, parts, name, f"{e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call (launch-to-launch)")
