"""GPU diagnostic: in-kernel stamps of the group-streamed S2+S3 kernel (instance_graph_stream_kernel) on the bench shape:
slot 0 start, 1 first barrier passed (wave 0), w >= 2: wave w done.  python tools/diag_stream.py"""
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
    run = lambda: sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False, zero_padding=False)
    for _ in range(20): run()
    torch.cuda.synchronize()
    st = torch.zeros(bench.B * 16, dtype=torch.int64, device=dev)
    lib.sn_debug_set_graph_stamps(st.data_ptr())
    g = run()
    torch.cuda.synchronize()
    lib.sn_debug_set_graph_stamps(None)
s = st.view(bench.B, 16).cpu().double()
t0 = s[:, 0].min()
q = lambda v: "median %.0f  p10 %.0f  p90 %.0f  max %.0f" % (v.median(), v.quantile(0.1), v.quantile(0.9), v.max())
print("start skew over images       " + q(s[:, 0] - t0))
print("start -> barrier (wave 0)    " + q(s[:, 1] - s[:, 0]))
print("  sorter: words + cls softmax  " + q(s[:, 2] - s[:, 0]))
print("  sorter: sort + records       " + q(s[:, 3] - s[:, 2]))
ends = s[:, 4:16] - s[:, 0:1]
print("start -> wave done (w 4..15) " + q(ends.flatten()))
print("  slowest wave of an image   " + q(ends.max(dim=1).values))
print("  fastest wave of an image   " + q(ends.min(dim=1).values))
print("  sorting wave (15)          " + q(ends[:, 11]))

print("vertices per image: median %d max %d" % (g["n"].float().median().item(), g["n"].max().item()))
