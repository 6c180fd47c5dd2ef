"""GPU diagnostic: Lloyd iteration time of discretization.kmeans at the bench shape (50,176 tokens, D=384, k=512)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import numpy as np, torch
import bench
from cpp_extension import ops
from discretization import kmeans as km
dev = torch.device("cuda", 0)
tokens, codebook, _ = bench.make_inputs(0, dev)
x = tokens[:, 1:, :].reshape(-1, bench.D).contiguous()
guess = x[torch.from_numpy(np.random.default_rng(0).choice(x.shape[0], bench.M, replace=False)).to(dev)]
for _ in range(2):
    book, avg, it = km.lloyd(x, guess, 1e-5, max_iter=5)
torch.cuda.synchronize(); t0 = time.perf_counter()
book, avg, it = km.lloyd(x, guess, 1e-5, max_iter=20)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"GPU: {it} Lloyd iterations in {dt*1e3:.1f} ms = {dt/it*1e3:.2f} ms per iteration ({x.shape[0]} tokens, k={bench.M}, D={bench.D}); avg distance {avg:.4f}")
ids = ops.assign_words(x[None], *ops.PackedCodebook().get(guess))[0]
for name, fn in (("update (stable sort + grouped sums)", lambda: ops.kmeans_update(x[None], ids[None], bench.M, sorted_route=True)),
                 ("update (every workgroup walks the ids)", lambda: ops.kmeans_update(x[None], ids[None], bench.M, sorted_route=False)), ("distances", lambda: ops.kmeans_distances(x[None], ids[None], guess))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(f"  {name}: {(time.perf_counter()-t0)/10*1e6:.0f} us")
from scipy.cluster.vq import kmeans as sk
xs = x[:12544].cpu().numpy(); gs = guess.cpu().numpy()
t0 = time.perf_counter(); sk(xs, gs, thresh=1e9); dt = time.perf_counter() - t0        # thresh huge: exactly two iterations
print(f"SciPy (reference path) on the host: 2 iterations over 12,544 tokens in {dt:.2f} s = {dt/2*4*1e3:.0f} ms per iteration at 50,176 tokens (1 thread)")
