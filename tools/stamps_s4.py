"""GPU diagnostic: in-kernel stamps of the one-round K-outer screen (variant 4) at the bench shape: medians over the waves of the
phase lengths in shader-clock cycles (s_memtime).  python tools/stamps_s4.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
lib = N.load()
lib.sn_assign_set_variant(4)
cb, packed = ops.PackedCodebook().get(codebook)
x = tokens[:, 1:, :]
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
def run():
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
for _ in range(3): run()
torch.cuda.synchronize()
st = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
lib.sn_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.sn_debug_set_stamps.restype = None
lib.sn_debug_set_stamps(st.data_ptr())
run(); torch.cuda.synchronize()
lib.sn_debug_set_stamps(None)
s = st.view(1024, 16)[:, :5].double()
t0 = s[:, 0].min()
names = ["start (after the first wave)", "prologue", "main loop", "keys", "merge + outputs"]
print("start skew: median %.0f max %.0f" % ((s[:, 0] - t0).median(), (s[:, 0] - t0).max()))
for i in range(1, 5):
    d = s[:, i] - s[:, i - 1]
    print("%-18s median %7.0f  min %7.0f  max %7.0f" % (names[i], d.median(), d.min(), d.max()))
print("whole wave         median %7.0f  max %7.0f; last end - first start %.0f" % ((s[:, 4] - s[:, 0]).median(), (s[:, 4] - s[:, 0]).max(), s[:, 4].max() - t0))
