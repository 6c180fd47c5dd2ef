"""Host cost of replaying a captured hipGraph: graphs of N tiny kernels (negligible GPU work), replayed back to back on one
stream and round-robin on four - microseconds per replay.  Tells whether the replayed bench is bound by the launch path."""
import sys, time
import torch
sys.path.insert(0, "schemanet-pytorch_amd")
dev = torch.device("cuda", 0)
x = torch.zeros(64, device=dev)
for n_nodes in (1, 4, 16, 32):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n_nodes):
            x.add_(1.0)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    t_submit = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    streams = [torch.cuda.Stream() for _ in range(4)]
    gs = []
    for k in range(4):
        gk = torch.cuda.CUDAGraph()
        xk = torch.zeros(64, device=dev)
        with torch.cuda.graph(gk):
            for _ in range(n_nodes):
                xk.add_(1.0)
        gs.append(gk)
    for k in range(4):
        with torch.cuda.stream(streams[k]):
            gs[k].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(500):
        with torch.cuda.stream(streams[i % 4]):
            gs[i % 4].replay()
    t_submit4 = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all4 = time.perf_counter() - t0
    print(f"{n_nodes:3d} kernel nodes: one stream {t_submit / 500 * 1e6:7.1f} us per replay to submit, {t_all / 500 * 1e6:7.1f} us to finish; "
          f"four streams {t_submit4 / 500 * 1e6:7.1f} / {t_all4 / 500 * 1e6:7.1f}", flush=True)
