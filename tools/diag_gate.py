"""GPU diagnostic: per-CU placement and phase times of the token-stationary S1 screen with the token-phase gate
(in-kernel stamps).  python tools/diag_gate.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench

dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
cb, packed = ops.PackedCodebook().get(codebook)
lib = N.load()
x = tokens[:, 1:, :]
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
lib.sn_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.sn_debug_set_stamps.restype = None
n_waves = 4 * max((n_tok + 127) // 128, 2 * torch.cuda.get_device_properties(dev).multi_processor_count)
st = torch.zeros(n_waves * 16, dtype=torch.int64, device=dev)
lib.sn_debug_set_stamps(st.data_ptr())
for _ in range(3):
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)))
torch.cuda.synchronize()
lib.sn_debug_set_stamps(None)
s = st.view(n_waves, 16).cpu().double()
wg_ok = (s[0::4, 0] > 0).repeat_interleave(4)      # workgroups that ran
s = s[wg_ok]
n_waves = s.shape[0]
idle = s[:, 1] == 0                                 # waves without tokens never stamp slot 1
t0 = s[:, 0].min()
print("kernel span %.0f ticks; start skew %.0f" % (s[:, 3].max() - t0, s[:, 0].max() - t0))
w0 = s[0::4]                         # wave 0 of every workgroup carries arrival / CU slot
arr, slot = w0[:, 6].long(), w0[:, 7].long()
print("distinct CU slots %d; workgroups per slot histogram %s; arrival histogram %s" % (
    slot.unique().numel(), torch.bincount(torch.bincount(slot)).tolist(), torch.bincount(arr).tolist()))
per_slot = torch.bincount(slot)
for a in sorted(arr.unique().tolist()):
    sel = (arr == a).repeat_interleave(4) & ~idle
    q = s[sel]
    f = lambda v: "med %.0f max %.0f" % (v.median(), v.max())
    print(f"arrival {a}: {int(sel.sum()) // 4} workgroups; start {f(q[:, 0] - t0)}; gate passed at {f(q[:, 8] - t0)}; tokens done at {f(q[:, 1] - t0)}; "
          f"main loop {f(q[:, 2] - q[:, 1])}; end at {f(q[:, 3] - t0)}")
alone = (per_slot[slot] == 1).repeat_interleave(4) & ~idle
if alone.any():
    q = s[alone]
    print("workgroups alone on their CU: %d; main loop med %.0f max %.0f; end at med %.0f" % (int(alone.sum()) // 4, (q[:, 2] - q[:, 1]).median(), (q[:, 2] - q[:, 1]).max(), (q[:, 3] - t0).median()))
rt = (s[:, 10] - s[:, 9])[~idle]          # 100 MHz ticks
mt = (s[:, 3] - s[:, 0])[~idle]
print("waves without tokens: %d of %d" % (int(idle.sum()), n_waves))
print("s_memtime ticks per 10 ns (s_memrealtime): median %.2f  (=> s_memtime runs at %.0f MHz); kernel span by realtime: %.1f us" % (
    (mt / rt).median(), (mt / rt).median() * 100, (s[~idle][:, 10].max() - s[:, 9].min()) / 100.0))
