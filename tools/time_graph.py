"""GPU diagnostic: instance_graph_kernel time on the bench shape (HIP events inside the library), warm clocks."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench
from cpp_extension import _native as N
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
lib = N.load()
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    ing = disc.assign(tokens[:, 1:, :])
    zp = os.environ.get("SN_ZERO_PADDING", "1") != "0"
    run = lambda: sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False, zero_padding=zp)
    for _ in range(100): run()
    torch.cuda.synchronize()
    lib.sn_profile_enable(100)
    for _ in range(100): run()
    torch.cuda.synchronize()
n = lib.sn_profile_count(2); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(2, buf, n)
v = sorted(buf)
print("instance_graph_kernel: median %.1f us  p10 %.1f  p90 %.1f (n=%d)" % (v[n // 2] * 1e3, v[n // 10] * 1e3, v[9 * n // 10] * 1e3, n))

# ---- the fused head-mean input: raw per-head logits [B * 6, L + 1, L + 1], head mean inside the kernel
H = int(os.environ.get("SN_HEADS", "6"))
g = torch.Generator().manual_seed(103)
ext = torch.randn(bench.B * H, bench.L + 1, bench.L + 1, generator=g).to(dev)
heads = ext.reshape(bench.B, H, bench.L + 1, bench.L + 1)
with torch.no_grad():
    run_h = lambda: sn.instance_graph_padded(ing, heads[:, :, 1:, 1:], heads[:, :, 0, 1:], mutate_inputs=False, zero_padding=False)
    for _ in range(20): run_h()
    torch.cuda.synchronize()
    lib.sn_profile_enable(50)
    for _ in range(50): run_h()
    torch.cuda.synchronize()
n = lib.sn_profile_count(2); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(2, buf, n)
v = sorted(buf)
byts = bench.B * H * bench.L * bench.L * 4
print("instance_graph_kernel, %d heads: median %.1f us  p10 %.1f  p90 %.1f  (attention read %.0f MB = %.2f TB/s)" % (
    H, v[n // 2] * 1e3, v[n // 10] * 1e3, v[9 * n // 10] * 1e3, byts / 1e6, byts / (v[n // 2] * 1e-3) / 1e12))
