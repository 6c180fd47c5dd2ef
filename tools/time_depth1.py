import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/schemanet-pytorch_amd")
import torch, bench
from schema_inference.utils.graph_replay import GraphedStep
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    g = GraphedStep(lambda: bench.step(disc, sn, m, tokens, attn, side_stream=True))
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): g.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("priority", os.environ.get("SN_CLASS_STREAM_PRIORITY", "0"), "depth1 %.1f us per step = %.0f k img/s" % (1e6 * dt / 300, 256 * 300 / dt / 1e3))
