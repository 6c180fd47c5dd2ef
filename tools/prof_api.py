"""GPU diagnostic: `SchemaNetPredictor.forward` called N times on resident head-averaged taps (bench.api_leg's third case) - run
under `rocprofv3 --kernel-trace --stats` (from /tmp) to see what the API route launches beside the hand-written step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import bench
from bench import B, D, H, L, K
import discretization
import schema_inference.graph as graph
from schema_inference.utils import IngredientModelWrapper
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
disc, sn, m = bench.make_model(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(bench.make_codebook(dev))
g = lambda s_: torch.Generator().manual_seed(s_)  # noqa: E731
batches = []
for i in range(4):
    mid = torch.randn(L + 1, B, D, generator=g(100 + 10 * i)).to(dev)
    ext = torch.randn(B * H, L + 1, L + 1, generator=g(103 + 10 * i)).to(dev)
    if heads == 1:
        ext = ext.reshape(B, H, L + 1, L + 1).mean(dim=1).contiguous()
    batches.append((mid, ext))
wrapper = IngredientModelWrapper(bench._ResidentBackbone(batches), discretization.DiscretizationModule(disc))
pred = graph.SchemaNetPredictor(wrapper, sn, m).eval()
pred.matcher.cache_atlas = False
x = torch.empty(B, 3, 1, 1, device=dev)
with torch.no_grad():
    for _ in range(8):
        pred(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        last = pred(x)["pred"]
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("heads", heads, "img/s %.0f" % (B * n / dt), "us per call %.1f" % (1e6 * dt / n), "host us per call %.1f" % (1e6 * t_host / n), "graphs", len(pred._graphs))
