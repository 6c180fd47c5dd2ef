"""GPU diagnostic: eager steps of the bench shape on a PRUNED atlas (70 % of the class vertices under the threshold), compacted
or not (SN_ATLAS_COMPACT), for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench
import schema_inference.graph as graph
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
K, M = bench.K, bench.M
g = torch.Generator().manual_seed(44)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    low = (torch.rand(K, M, generator=g) < 0.7).to(dev)
    sn.vertex_weights.tensor.copy_(torch.where(low, sn.vertex_weights.tensor * 1.0e-4, sn.vertex_weights.tensor))
    run = lambda: bench.step(disc, sn, m, tokens, attn, class_branch_first=False, side_stream=False)      # noqa: E731
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n): last = run()
    torch.cuda.synchronize()
print("compact", os.environ.get("SN_ATLAS_COMPACT", "1"), "pruned flag", sn._atlas_compaction_pays(), "%.1f us per eager step" % (1e6 * (time.perf_counter() - t0) / n))
