"""Layer 1's product with and without layer 2's Linear in its epilogue (sn_gemm_args.next_w_*), class-graph and instance
shapes of the bench: event-pair time per launch and the in-kernel stamps (loop / epilogue cycles per workgroup)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
from cpp_extension import ops, _native as N

dev = torch.device("cuda", 0)
lib = N.load()
lib.sn_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]; lib.sn_debug_set_gemm_stamps.restype = None
g = torch.Generator().manual_seed(1)
for (G, n, nv_hi) in ((100, 512, 512), (256, 196, 125)):
    nv = torch.randint(nv_hi * 3 // 4, nv_hi + 1, (G,), generator=g, dtype=torch.int32).to(dev)
    ext = nv.max().reshape(1).to(torch.int32)
    e = (torch.rand(G, n, n, generator=g) / n).to(dev)
    adj = ops.gcn_adjacency_planes(e, extent=ext, n_valid=nv)
    table = torch.randn(513, 256, generator=g).to(dev); table[512] = 0
    ids = torch.randint(0, 512, (G, n), generator=g).to(dev)
    t_hi, t_lo = ops.table_planes(table)
    W2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
    w2n, w2p = ops.next_layer_weight_planes(W2), ops.split_planes(W2)
    gam, bet, bias = torch.ones(256, device=dev), torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    per_graph = os.environ.get("SN_TIME_PER_GRAPH", "1") == "1" and n < 256          # (instance graphs take their extents per graph: GNN._forward_mfma)
    if per_graph:
        adj = ops.gcn_adjacency_planes(e, extent=ext, n_valid=nv.contiguous(), per_graph=True)
    gext = nv.contiguous() if per_graph else ext
    kw = dict(bias=bias, layernorm=(gam, bet, 1e-5), relu=True, rows_valid=nv, m_extent=gext, k_extent=gext, b_table=(t_hi, t_lo, ids))
    forms = {"fused": lambda: ops.gcn_gemm(adj, None, G, want_planes=n, next_w=w2n, **kw),
             "H1 planes": lambda: ops.gcn_gemm(adj, None, G, want_planes=256, **kw)}
    h1 = forms["H1 planes"]()["planes"]
    forms["stand-alone Linear"] = lambda: ops.gcn_gemm(w2p, h1, G, want_planes=n)
    zt2 = forms["fused"]()["planes"]
    nodes = torch.rand(G, n, generator=g).to(dev)
    kw_pool = {k_: v for k_, v in kw.items() if k_ != "b_table"}
    forms["pooled"] = lambda: ops.gcn_gemm(adj, zt2, G, pool_w=nodes, **kw_pool)
    for name, fn in forms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        blocks = 8 * ((G + 7) // 8) * 4
        st = torch.zeros(blocks * 4 * 16 * 2, dtype=torch.int64, device=dev)
        lib.sn_debug_set_gemm_stamps(st.data_ptr())
        fn(); torch.cuda.synchronize()
        lib.sn_debug_set_gemm_stamps(None)
        s8 = st.view(-1, 16).cpu().double(); s8 = s8[s8[:, 0] > 0]
        msg = ""
        if len(s8):
            t0 = s8[:, 0].min()
            msg = (f"span {s8[:, 2].max() - t0:.0f} cycles, loop median {(s8[:, 1] - s8[:, 0]).median():.0f}, epilogue median "
                   f"{(s8[:, 2] - s8[:, 1]).median():.0f} max {(s8[:, 2] - s8[:, 1]).max():.0f}, waves {len(s8)}")
        if name == "fused" and len(s8):
            d = lambda a, b: (s8[:, a] - s8[:, b]).median().item()
            msg += (f"; LayerNorm {d(3, 1):.0f}, fragments + barriers {d(4, 3):.0f}, 16 k-steps {d(5, 4):.0f}, barrier {d(6, 5):.0f}, "
                    f"stores of half 0 + all of half 1 {d(2, 6):.0f}")
        if name != "fused" and len(s8):
            msg += f"; in the loop: wait + barrier median {s8[:, 3].median():.0f}, copy issue median {s8[:, 4].median():.0f} (cycles per wave)"
        if len(s8) and s8[:, 8].min() > 0:
            r0 = s8[:, 8].min()
            msg += (f"; wall: first entry -> last end {(s8[:, 9].max() - r0) / 100:.1f} us, entries spread over {(s8[:, 8].max() - r0) / 100:.1f} us, median "
                    f"workgroup {((s8[:, 9] - s8[:, 8]).median()) / 100:.1f} us, workgroups {len(s8) // 4}")
        if len(s8) and s8[:, 8].min() > 0 and os.environ.get("SN_TIME_HIST") == "1":
            ent = ((s8[:, 8] - s8[:, 8].min()) / 100)[::4]
            hist = torch.histc(ent.float(), bins=12, min=0, max=float(ent.max()) + 1e-3)
            msg += "; entries per " + f"{float(ent.max()) / 12:.1f}" + " us bin: " + " ".join(str(int(v)) for v in hist.tolist())
        print(f"G={G} n={n} {name:20s} {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us per launch (launch-to-launch); {msg}", flush=True)
