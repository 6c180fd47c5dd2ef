"""Round 6: sn_gcn_gemm with 128-row tiles (two 4-wave workgroups per CU) against 256-row tiles (one 8-wave workgroup per CU, waves 4-7
staggered or not) on the class-side products of the bench (K = 100 graphs of 512 vertices, E = 256), of config [4] (101 x 1024) and of
config [3] (wide, plain product): bit equality of every output, event-pair time per launch and the in-kernel stamps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N

dev = torch.device("cuda", 0)
lib = N.load()
g = torch.Generator().manual_seed(1)
FORMS = (("128", 128, 1), ("256 staggered", 256, 1), ("256 in step", 256, 0))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def stamps(fn, waves_total):
    st = torch.zeros(waves_total * 16, dtype=torch.int64, device=dev)
    lib.sn_debug_set_gemm_stamps(st.data_ptr())
    fn(); torch.cuda.synchronize()
    lib.sn_debug_set_gemm_stamps(None)
    s8 = st.view(-1, 16).cpu().double()
    return s8[s8[:, 0] > 0]


def flat(out):
    r = []
    for k in sorted(out):
        v = out[k]
        if v is None:
            continue
        if torch.is_tensor(v):
            r.append(v)
        else:                                   # Planes
            r += [v.hi, v.lo]
    return r


def run(G, n, label, which=("fused", "pooled", "plain")):
    nv = torch.randint(n * 3 // 4, n + 1, (G,), generator=g, dtype=torch.int32).to(dev)
    ext = nv.max().reshape(1).to(torch.int32)
    e = (torch.rand(G, n, n, generator=g) / n).to(dev)
    adj = ops.gcn_adjacency_planes(e, extent=ext, n_valid=nv)
    table = torch.randn(513, 256, generator=g).to(dev); table[512] = 0
    ids = torch.randint(0, 512, (G, n), generator=g).to(dev)
    t_hi, t_lo = ops.table_planes(table)
    W2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
    w2n = ops.next_layer_weight_planes(W2)
    gam, bet, bias = torch.rand(256, generator=g).to(dev) + 0.5, torch.randn(256, generator=g).to(dev) * 0.1, torch.randn(256, generator=g).to(dev) * 0.1
    nodes = torch.rand(G, n, generator=g).to(dev)
    kw = dict(bias=bias, layernorm=(gam, bet, 1e-5), relu=True, rows_valid=nv, m_extent=ext, k_extent=ext)
    lib.sn_debug_set_gemm_tile(128, 1)
    zt2 = ops.gcn_gemm(adj, None, G, want_planes=n, next_w=w2n, b_table=(t_hi, t_lo, ids), **kw)["planes"]
    bt = ops.split_planes(torch.randn(G, 256, n, generator=g).to(dev))
    forms = {
        "fused": lambda: ops.gcn_gemm(adj, None, G, want_planes=n, next_w=w2n, b_table=(t_hi, t_lo, ids), **kw),
        "pooled": lambda: ops.gcn_gemm(adj, zt2, G, pool_w=nodes, **kw),
        "plain": lambda: ops.gcn_gemm(adj, bt, G, want_c=True, want_planes=256),
        "ln planes": lambda: ops.gcn_gemm(adj, bt, G, want_planes=256, **kw),
    }
    for name in which:
        fn = forms[name]
        ref = None
        for fl, tm, stg in FORMS:
            lib.sn_debug_set_gemm_tile(tm, stg)
            out = flat(fn()); torch.cuda.synchronize()
            if ref is None:
                ref = [t.clone() for t in out]
                same = "reference"
            else:
                same = "bit-identical" if all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(ref, out)) else \
                       "DIFFERENT: max |d| " + ", ".join(f"{(a.float() - b.float()).abs().max().item():.3e}" for a, b in zip(ref, out))
            us = timed(fn)
            waves = 8 * ((G + 7) // 8) * ((n + tm - 1) // tm) * (tm // 32)
            s8 = stamps(fn, waves)
            msg = ""
            if len(s8):
                t0 = s8[:, 0].min()
                msg = (f"span {s8[:, 2].max() - t0:.0f} cycles, loop median {(s8[:, 1] - s8[:, 0]).median():.0f}, epilogue median "
                       f"{(s8[:, 2] - s8[:, 1]).median():.0f} max {(s8[:, 2] - s8[:, 1]).max():.0f}, waves {len(s8)}")
                if s8[:, 8].min() > 0:      # (the 100 MHz clock all XCDs share: entry of the first workgroup .. end of the last)
                    r0 = s8[:, 8].min()
                    msg += (f"; wall (100 MHz clock) first entry -> last end {(s8[:, 9].max() - r0) / 100:.1f} us, entries spread over "
                            f"{(s8[:, 8].max() - r0) / 100:.1f} us, median workgroup {((s8[:, 9] - s8[:, 8]).median()) / 100:.1f} us")
            print(f"{label} {name:10s} tile {fl:14s} {us:7.1f} us per launch; {same}; {msg}", flush=True)
    lib.sn_debug_set_gemm_tile(0, 1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "c2":          # (lab builds: SN_LIB_PATH=tools/lab/bin/<name>.so python tools/time_gemm_tile.py c2)
        print("library:", os.environ.get("SN_LIB_PATH", "default"))
        run(100, 512, "C2", which=("fused", "pooled"))
        sys.exit(0)
    run(100, 512, "C2 class graphs (100 x 512)")
    run(101, 1024, "config [4] class graphs (101 x 1024)", which=("fused", "pooled"))
    run(37, 300, "odd shape (37 x 300)", which=("fused", "pooled", "plain", "ln planes"))
    run(1000, 512, "1000 graphs x 512", which=("plain",))
