"""GPU diagnostic: in-kernel stamps (s_memtime ticks = shader cycles) of the one-round K-outer screen with eight waves per CU
(assign_screen5_kernel) on the bench shape.  Slots: 0 start, 1 prologue barrier, 5 chunk 0 in LDS, 6 last chunk in LDS,
2 main loop done, 3 keys done, 7 first merge barrier passed, 8 second, 4 records written.  python tools/diag_s5.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench

dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
lib = N.load()
cb, packed = ops.PackedCodebook().get(codebook)
assert lib.sn_assign_set_variant(5) == 0
x = tokens[:, 1:, :]
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
mode = int(os.environ.get("S5_MODE", "2"))          # 2: the screen only


def run():
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), mode, N.stream_ptr(dev)), "assign")


for _ in range(5):
    run()
torch.cuda.synchronize()
n_waves = 8 * 256
st = torch.zeros(n_waves * 16, dtype=torch.int64, device=dev)
lib.sn_debug_set_stamps(st.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.sn_debug_set_stamps(None)
s = st.view(n_waves, 16).cpu().double()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
q = lambda v: "median %.0f  p10 %.0f  p90 %.0f  max %.0f" % (v.median(), v.quantile(0.1), v.quantile(0.9), v.max())
print("waves stamped", s.shape[0])
print("kernel span (first start -> last end) %.0f cycles; start skew %.0f" % (s[:, 4].max() - t0, s[:, 0].max() - t0))
print("  0->1 prologue (hcs, first copies issued)   " + q(s[:, 1] - s[:, 0]))
print("  1->5 chunk 0 landed                        " + q(s[:, 5] - s[:, 1]))
print("  5->6 chunks 0 .. n-2 multiplied, last in   " + q(s[:, 6] - s[:, 5]))
print("  6->2 last chunk multiplied                 " + q(s[:, 2] - s[:, 6]))
print("  2->3 keys                                  " + q(s[:, 3] - s[:, 2]))
print("  3->7 windows, ds_min, barrier              " + q(s[:, 7] - s[:, 3]))
print("  7->8 window test, atomic or, barrier       " + q(s[:, 8] - s[:, 7]))
print("  8->4 records                               " + q(s[:, 4] - s[:, 8]))
print("  0->4 wave lifetime                         " + q(s[:, 4] - s[:, 0]))
import torch as _t
wv = _t.arange(s.shape[0]) % 8
for name, sel in (("sg=0 waves", wv < 4), ("sg=1 waves", wv >= 4)):
    z = s[sel]
    print("  chunk 5, %s: S0 S1          %s" % (name, q(z[:, 10] - z[:, 9])))
    print("           S2 (+ leftover addresses)  " + q(z[:, 11] - z[:, 10]))
    print("           L S3 S4                    " + q(z[:, 12] - z[:, 11]))
    print("           wait + barrier             " + q(z[:, 13] - z[:, 12]))
    print("           copies (sg=0) + S5         " + q(z[:, 14] - z[:, 13]))
    print("           whole chunk                " + q(z[:, 14] - z[:, 9]))
# timed by events, screen only and with the stand-alone finish
for m in (2, 0):
    mode = m
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.sn_profile_enable(40)
    for _ in range(40):
        run()
    torch.cuda.synchronize()
    n = lib.sn_profile_count(0); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(0, buf, n)
    v = sorted(buf)
    print("mode %d: screen by the library's event pair: median %.1f us, min %.1f us" % (m, v[n // 2] * 1e3, v[0] * 1e3))
    lib.sn_profile_enable(0)
lib.sn_assign_set_variant(0)
