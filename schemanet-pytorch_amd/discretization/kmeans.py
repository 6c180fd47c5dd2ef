"""Codebook extraction on the GPU: Lloyd's k-means with the S1 assignment kernel as the E-step.

Replaces the clustering step of the reference's `scripts/extract_ingredients.py:28-56,117-124`
(`KMeansClustering("cpu_kmeans")` = `scipy.cluster.vq.kmeans(x, num_clusters)`), i.e. the producer of
the `cluster_<M>_from_<N>.pth` file that `Discretization.initial_vocabulary` loads.  The algorithm is
SciPy's (scipy/cluster/vq.py `kmeans`, `_kmeans`, `_kpoints`; `_vq.update_cluster_means`), float32 path:

    repeat:  codes  = nearest centre of every observation              (sn_assign_words: exact, first index on ties)
             avg    = mean_t |x_t - centre_code(t)|                      (sn_kmeans_distances, fp64)
             centre = fp32 sum of the members in observation order / n   (sn_kmeans_update)
             drop centres without members
    until |avg_previous - avg| <= thresh;   k given as a number: `iter` restarts from k random distinct
    observations, keep the book with the lowest distortion.

Observations can be sharded over ranks (every rank holds its own rows, the same initial centres): the
per-centre sums, counts and distances are added with one all-reduce per iteration, so every rank ends
with the same book.  On one GPU the book is bit-identical to SciPy's whenever SciPy's fp32 argmin agrees
with the exact one (tests/test_gpu_parity.py::test_kmeans_*).
"""
from typing import Callable, Optional, Tuple, Union

import numpy as np
import torch


def _hip_backend():
    from cpp_extension import ops

    def assign(x, centres):
        cb, packed = ops.PackedCodebook().get(centres)
        return ops.assign_words(x[None], cb, packed)[0]

    def update(x, ids, k):
        return ops.kmeans_update(x[None], ids[None], k)

    def distances(x, ids, centres):
        return ops.kmeans_distances(x[None], ids[None], centres)

    return assign, update, distances


def _all_reduce(t: torch.Tensor, group) -> torch.Tensor:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t


def lloyd(obs: torch.Tensor, guess: torch.Tensor, thresh: float = 1e-5, group=None,
          backend: Optional[Tuple[Callable, Callable, Callable]] = None, max_iter: int = 10_000
          ) -> Tuple[torch.Tensor, float, int]:
    """SciPy's `_kmeans(obs, guess, thresh)`.  obs [N_local, D] f32 (this rank's observations), guess
    [k, D] f32 (same on every rank).  -> (code book [k' <= k, D] f32, average distance, iterations)."""
    assign, update, distances = backend or _hip_backend()
    book = guess.to(torch.float32).contiguous()
    prev = [float("inf")]
    diff, it = float("inf"), 0
    while diff > thresh and it < max_iter:
        ids = assign(obs, book)
        dist = distances(obs, ids, book)
        acc = torch.stack([dist.sum(), torch.tensor(float(dist.numel()), dtype=torch.float64, device=dist.device)])
        acc = _all_reduce(acc, group)
        prev = (prev + [float(acc[0] / acc[1])])[-2:]
        sums, counts = update(obs, ids, book.shape[0])
        _all_reduce(sums, group)
        _all_reduce(counts, group)
        has = counts > 0
        book = (sums[has] / counts[has].to(torch.float32)[:, None]).contiguous()
        diff = abs(prev[0] - prev[1])
        it += 1
    return book, prev[1], it


def kmeans(obs: torch.Tensor, k_or_guess: Union[int, torch.Tensor], iter: int = 20, thresh: float = 1e-5,
           rng: Union[None, int, np.random.Generator, np.random.RandomState] = None, group=None,
           backend=None) -> Tuple[torch.Tensor, float]:
    """`scipy.cluster.vq.kmeans(obs, k_or_guess, iter, thresh, rng=rng)` on GPU tensors.  With a number
    of clusters the initial books are rows `rng.choice(N, k, replace=False)` of `obs` (SciPy's `_kpoints`);
    sharded runs must pass an explicit guess (the draw is over local rows)."""
    if torch.is_tensor(k_or_guess) and k_or_guess.numel() != 1:
        book, dist, _ = lloyd(obs, k_or_guess.to(obs.device), thresh, group, backend)
        return book, dist
    k = int(k_or_guess)
    if k < 1:
        raise ValueError("Asked for %d clusters." % k)
    if iter < 1:
        raise ValueError(f"iter must be at least 1, got {iter}")
    if not isinstance(rng, (np.random.Generator, np.random.RandomState)):
        rng = np.random.default_rng(rng)
    best, best_dist = None, float("inf")
    for _ in range(iter):
        idx = torch.from_numpy(np.asarray(rng.choice(obs.shape[0], size=k, replace=False))).to(obs.device)
        book, dist, _ = lloyd(obs, obs[idx], thresh, group, backend)
        if dist < best_dist:
            best, best_dist = book, dist
    return best, best_dist


class KMeansClustering:
    """reference scripts/extract_ingredients.py:28-56: `KMeansClustering(num_clusters, method)(x) -> centres`.
    method "hip_kmeans" (numpy or tensor in, float32 numpy out like the reference's methods)."""

    def __init__(self, num_clusters: int, method: str = "hip_kmeans", iter: int = 20, rng=None):
        if method != "hip_kmeans":
            raise ValueError(f"unknown method {method!r} (this package provides 'hip_kmeans')")
        self.num_clusters, self.method, self.iter, self.rng = num_clusters, method, iter, rng

    def __call__(self, x) -> np.ndarray:
        if not torch.cuda.is_available():
            raise RuntimeError("hip_kmeans needs a GPU (the HIP path has no CPU fallback)")
        t = torch.as_tensor(x, dtype=torch.float32).cuda()
        centres, _ = kmeans(t, self.num_clusters, iter=self.iter, rng=self.rng)
        return centres.cpu().numpy()
