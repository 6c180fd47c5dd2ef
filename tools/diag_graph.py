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
    zp = os.environ.get("SN_ZERO_PADDING", "0") != "0"        # (0: what the predictor launches - the compile-time configuration of the kernel)
    for _ in range(3): g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False, zero_padding=zp)
    st = torch.zeros(bench.B * 16, dtype=torch.int64, device=dev)
    lib.sn_debug_set_graph_stamps(st.data_ptr())
    g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False, zero_padding=zp)
    torch.cuda.synchronize(); lib.sn_debug_set_graph_stamps(None)
s = st.view(bench.B, 16).cpu().double()
names = ["attn rows -> LDS, cls softmax, sync", "group positions", "vertices out", "row map", "edges (cells + store)"]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print("%-40s median %8.0f  max %8.0f cycles" % (nm, d.median(), d.max()))
print("wave 0 of each block: pass a (column sums) median %.0f max %.0f; pass b (stage + gather) median %.0f max %.0f" % (s[:, 6].median(), s[:, 6].max(), s[:, 7].median(), s[:, 7].max()))
print("wave 0 of each block: before the row loop median %.0f; normalise + store median %.0f max %.0f" % (s[:, 9].median(), s[:, 8].median(), s[:, 8].max()))
print("before the first barrier: the sorting wave is done after median %.0f max %.0f cycles, wave 0 (rows) after median %.0f max %.0f" % (
    (s[:, 10] - s[:, 0]).median(), (s[:, 10] - s[:, 0]).max(), (s[:, 11] - s[:, 0]).median(), (s[:, 11] - s[:, 0]).max()))
print("   wave 3 (row wave on the sorting wave's SIMD) after median %.0f max %.0f, wave 5 after median %.0f max %.0f" % (
    (s[:, 14] - s[:, 0]).median(), (s[:, 14] - s[:, 0]).max(), (s[:, 15] - s[:, 0]).median(), (s[:, 15] - s[:, 0]).max()))
print("sorting wave: cls soft-max done at median %.0f, positions grouped at %.0f, vertices written at %.0f (cycles from the start)" % (
    (s[:, 12] - s[:, 0]).median(), (s[:, 13] - s[:, 0]).median(), (s[:, 10] - s[:, 0]).median()))
print("total per block median %.0f max %.0f; n_i median %d" % ((s[:, 5] - s[:, 0]).median(), (s[:, 5] - s[:, 0]).max(), g["n"].float().median()))

