"""GPU diagnostic: what a plain streaming read reaches on this box (torch reductions / copies), for the roofline denominator."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (77, 256, 1024, 4096):
    x = torch.randn(mb * 1024 * 1024 // 4, device=dev)
    y = torch.empty_like(x)
    ts = t(lambda: x.sum()); tm = t(lambda: x.max()); tc = t(lambda: y.copy_(x))
    print(f"{mb} MB: sum {x.numel()*4/ts/1e9:.0f} GB/s ({ts*1e6:.1f} us)  max {x.numel()*4/tm/1e9:.0f} GB/s  copy {2*x.numel()*4/tc/1e9:.0f} GB/s (read+write)", flush=True)
