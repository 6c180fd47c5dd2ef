import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/schemanet-pytorch_amd")
import torch, bench
from schema_inference.utils.graph_replay import GraphedStep
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    for ss in (False, True):
        g = GraphedStep(lambda: bench.step(disc, sn, m, tokens, attn, side_stream=ss))
        a = g.replay().clone(); b = g.replay().clone()
        print("side_stream", ss, "memset nodes replaced / left", g.memsets_replaced, g.memsets_left, "replays equal", torch.equal(a, b))
