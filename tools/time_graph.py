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
