"""GPU diagnostic: A/B timing of the launch options of the token-stationary S1 screen inside ONE process (the boxes
differ by several per cent): gate x balance, interleaved rounds, HIP events inside the library.
python tools/ab_assign.py [rounds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N
import bench

dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tokens, codebook, attn = bench.make_inputs(0, dev)
lib = N.load()
lib.sn_assign_set_variant(0)
cb, packed = ops.PackedCodebook().get(codebook)
x = tokens[:, 1:, :]
n_tok = x.shape[0] * x.shape[1]
ws = torch.zeros(lib.sn_assign_workspace_bytes(n_tok), dtype=torch.uint8, device=dev)
out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int64, device=dev)
ref = None
def run():
    N.check(lib.sn_assign_words(N.ptr(x), x.shape[0], x.shape[1], x.stride(0), x.stride(1), N.ptr(cb), N.ptr(packed), bench.M, bench.D,
                                N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws.numel(), 0, N.stream_ptr(dev)), "assign")
configs = [(0, 0), (1, 0), (0, 1), (1, 1)]
times = {c: [] for c in configs}
for _ in range(200):          # warm the clocks
    run()
torch.cuda.synchronize()
for r in range(rounds):
    for c in configs:
        lib.sn_debug_set_assign_options(*c)
        run(); torch.cuda.synchronize()
        lib.sn_profile_enable(8)
        for _ in range(8):
            run()
        torch.cuda.synchronize()
        n = lib.sn_profile_count(0); buf = (C.c_float * n)(); lib.sn_profile_elapsed_ms(0, buf, n)
        times[c] += [v * 1e3 for v in buf]
        lib.sn_profile_enable(0)
        if ref is None: ref = out.clone()
        assert torch.equal(ref, out)
for c in configs:
    v = sorted(times[c])
    print("gate %d balance %d: screen median %.2f us  p10 %.2f  p90 %.2f  (n=%d)" % (c[0], c[1], v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10], len(v)), flush=True)
