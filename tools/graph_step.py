"""GPU experiment: the whole schema-inference step captured in one hipGraph (torch.cuda.CUDAGraph)
versus eager launches: same result, step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import bench
from cpp_extension import ops

dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
votes = torch.zeros(bench.K + 1, device=dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    for _ in range(5):
        pred_e = bench.step(disc, sn, m, tokens, attn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        pred_e = bench.step(disc, sn, m, tokens, attn)
        ops.class_votes_(pred_e, votes)
    torch.cuda.synchronize()
    print("eager: %.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            bench.step(disc, sn, m, tokens, attn)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        pred_g = bench.step(disc, sn, m, tokens, attn)
        ops.class_votes_(pred_g, votes)
    torch.cuda.synchronize()
    print("captured", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("max |graph - eager| =", float((pred_g - pred_e).abs().max()), "equal:", bool(torch.equal(pred_g, pred_e)), flush=True)
    for n in (30, 100):
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        print("graph replay x%d: %.3f ms/step" % (n, (time.perf_counter() - t0) / n * 1e3), flush=True)

# ---- two steps in flight: two captured graphs replayed alternately on two streams
from schema_inference.utils.graph_replay import GraphedStep
with torch.no_grad():
    def one():
        p_ = bench.step(disc, sn, m, tokens, attn)
        ops.class_votes_(p_, votes)
        return p_
    NG = int(os.environ.get('SN_NG', '2'))
    gs = [GraphedStep(one) for _ in range(NG)]
    streams = [torch.cuda.Stream() for _ in range(NG)]
    torch.cuda.synchronize()
    for n in (30, 100):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % NG]):
                gs[i % NG].replay()
        torch.cuda.synchronize()
        print(f"{NG} graphs on {NG} streams " "x%d: %.3f ms/step" % (n, (time.perf_counter() - t0) / n * 1e3), flush=True)
    print("equal:", bool(torch.equal(gs[0].outputs, pred_e)), bool(torch.equal(gs[1].outputs, pred_e)))
