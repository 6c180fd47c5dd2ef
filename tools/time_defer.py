"""GPU diagnostic: the instance-graph kernel with and without the deferred S1 finish (round 4) on the bench shape -
launch time (HIP events inside the library) and phase stamps (shader cycles)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench
from cpp_extension import _native as N
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
lib = N.load()
lib.sn_debug_set_graph_stamps.argtypes = [C.c_void_p]; lib.sn_debug_set_graph_stamps.restype = None


def timed(run, n=100, kid=2):
    for _ in range(20): run()
    torch.cuda.synchronize()
    lib.sn_profile_enable(n)
    for _ in range(n): run()
    torch.cuda.synchronize()
    k = lib.sn_profile_count(kid); buf = (C.c_float * k)(); lib.sn_profile_elapsed_ms(kid, buf, k)
    lib.sn_profile_enable(0)
    v = sorted(buf)
    return v[k // 2] * 1e3, v[k // 10] * 1e3, v[9 * k // 10] * 1e3


def stamps(run):
    st = torch.zeros(bench.B * 16, dtype=torch.int64, device=dev)
    lib.sn_debug_set_graph_stamps(st.data_ptr())
    run()
    torch.cuda.synchronize(); lib.sn_debug_set_graph_stamps(None)
    s = st.view(bench.B, 16).cpu().double()
    f = lambda a, b: "%6.0f / %6.0f" % ((s[:, a] - s[:, b]).median(), (s[:, a] - s[:, b]).max())
    return ("first barrier %s | sorter starts sorting %s, done %s | wave 0 rows in %s | block %s | slots 14, 15 (fused: wave 3's finish requests out / complete) %s, %s (median / max cycles)" %
            (f(1, 0), f(12, 0), f(10, 0), f(11, 0), f(5, 0), f(14, 0), f(15, 0)))


with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    for name, tok in (("randn", tokens),):
        x = tok[:, 1:, :]
        final = disc.assign(x)
        ids, h = disc.assign(x, defer=True)
        a1, a2 = attn[:, 1:, 1:], attn[:, 0, 1:]
        plain = lambda: sn.instance_graph_padded(final, a1, a2, mutate_inputs=False, zero_padding=False)

        def fused():
            h.done = False
            return sn.instance_graph_padded(ids, a1, a2, mutate_inputs=False, zero_padding=False, rerank=h)
        print(name, "plain  : median %.1f us p10 %.1f p90 %.1f" % timed(plain), flush=True)
        print(name, "fused  : median %.1f us p10 %.1f p90 %.1f" % timed(fused), flush=True)
        print(name, "plain  :", stamps(plain))
        print(name, "fused  :", stamps(fused))
        assert torch.equal(ids, final)
        # S1 alone: the screen (kernel id 0) and the stand-alone re-rank (kernel id 1: flagged + overflow kernels)
        t0 = timed(lambda: disc.assign(x), kid=0)
        t1 = timed(lambda: disc.assign(x), kid=1)
        print(name, "screen : median %.1f us p10 %.1f p90 %.1f; stand-alone re-rank: median %.1f us p10 %.1f p90 %.1f" % (t0 + t1))
