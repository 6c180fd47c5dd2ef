"""GPU diagnostic: distribution of the vertex counts n_i of the bench batch."""
import sys, os
sys.path.insert(0, '" + os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "'); sys.path.insert(0, '/root/repo/schemanet-pytorch_amd')
import torch, bench
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    ing = disc.assign(tokens[:, 1:, :])
    g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False)
    n = g["n"].float()
    print("n_i: min %d median %d max %d; n_max %d" % (n.min(), n.median(), n.max(), int(g["n_max"])))
