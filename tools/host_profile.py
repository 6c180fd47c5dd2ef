"""Where does the host time of one step go? (cProfile over 50 eager steps)"""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    for _ in range(5): bench.step(disc, sn, m, tokens, attn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): bench.step(disc, sn, m, tokens, attn)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host enqueue per step %.3f ms, total per step %.3f ms" % ((t1 - t0) * 20, (t2 - t0) * 20))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): bench.step(disc, sn, m, tokens, attn)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    # graph capture of the same step
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): bench.step(disc, sn, m, tokens, attn)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        pred = bench.step(disc, sn, m, tokens, attn)
    ref = bench.step(disc, sn, m, tokens, attn)
    g.replay(); torch.cuda.synchronize()
    print("graph replay == eager:", torch.equal(pred, ref))
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("graph replay per step %.3f ms -> %.0f img/s" % ((t2 - t0) * 20, 256 / ((t2 - t0) / 50)))
