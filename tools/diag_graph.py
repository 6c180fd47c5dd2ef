"""GPU diagnostic: phase breakdown (shader cycles) of instance_graph_kernel."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench
from cpp_extension import _native as N
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
lib = N.load()
lib.sn_debug_set_graph_stamps.argtypes = [ctypes.c_void_p]; lib.sn_debug_set_graph_stamps.restype = None
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    ing = disc.assign(tokens[:, 1:, :])
    for _ in range(3): g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False)
    st = torch.zeros(bench.B * 8, dtype=torch.int64, device=dev)
    lib.sn_debug_set_graph_stamps(st.data_ptr())
    g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False)
    torch.cuda.synchronize(); lib.sn_debug_set_graph_stamps(None)
s = st.view(bench.B, 8).cpu().double()
names = ["attn rows -> LDS, cls softmax, sync", "group positions", "vertices out", "row map", "edges (cells + store)"]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print("%-40s median %8.0f  max %8.0f cycles" % (nm, d.median(), d.max()))
print("wave 0 of each block: pass a (column sums) median %.0f max %.0f; pass b (stage + gather) median %.0f max %.0f" % (s[:, 6].median(), s[:, 6].max(), s[:, 7].median(), s[:, 7].max()))
print("total per block median %.0f max %.0f; n_i median %d" % ((s[:, 5] - s[:, 0]).median(), (s[:, 5] - s[:, 0]).max(), g["n"].float().median()))

