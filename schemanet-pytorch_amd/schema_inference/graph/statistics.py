"""Per-class schema statistics for IR-Atlas initialisation, sharded over images.

Restates the two passes of the reference's `scripts/init_schema_net.py` (init_class_vertices
:43-65, init_graph :19-40, main :110-124) as an accumulator object:

    pass 1   class_vertex_sum[label] += feat_to_full_vertices(...)        [K, M]
    pass 2   class_edge_sum[label]   += feat_to_limited_edges(...)        [K, n_max, n_max]

The reference runs them in one process on one GPU with a python `+=` per sample.  Here every
rank accumulates the images of its shard on device (sn_stats_accumulate: deterministic image
order) and ONE all-reduce(SUM) per pass over RCCL/xGMI merges the ranks; sums are linear, so
the result equals the single-process one up to fp32 re-association.

Message sizes: pass 1 is K*M + K floats (205 KB at K=100, M=512) -> a single fused all-reduce
(latency bound).  Pass 2 is K*n_max^2 + K floats (105 MB at n_max=512, 419 MB at 1024): issued as
reduce_scatter + all_gather on the flat buffer, so that on the fully connected xGMI mesh each rank
exchanges 1/W shards with every peer directly instead of pushing the whole buffer around a ring.
The flat buffers are allocated with `_SLACK` spare (zero) elements behind the logical end, so the
collective runs on the buffer rounded up to a multiple of the world size whatever K and n_max are
(K*n_max^2 + K is 26 214 500 at C2: not a multiple of 8) - no copy, no fallback to the ring.
`SchemaStatistics.last_collective` says which form the last merge took.
"""
import logging
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world_size: int) -> torch.Tensor:
    """Image indices of `rank`: r, r+W, r+2W, ...  -- the interleaved split of
    DistributedSampler (reference data/__init__.py:106-122)."""
    return torch.arange(min(rank, n_items), n_items, world_size)


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_SLACK = 256          # spare elements behind a flat buffer: any world size up to 256 divides some length in [n, n + 256)


class SchemaStatistics:
    """Accumulates per-class vertex / edge statistics for one rank and merges ranks."""

    large_bytes = 8 << 20      # flat buffers of at least this many bytes are merged by reduce_scatter + all_gather
    #: run the merge through the backend even in a world of one (tests: the RCCL calls themselves on a one-GPU box)
    merge_single_rank = False

    def __init__(self, num_classes: int, num_vertices: int, class_max_vertices: Optional[int] = None,
                 device: torch.device = None):
        self.K, self.M = num_classes, num_vertices
        self.n_max = class_max_vertices or num_vertices
        self.device = device
        f32 = dict(dtype=torch.float32, device=device)
        # one flat buffer per pass so a pass needs exactly one collective: [sums..., n_tracked]
        self._v_store = torch.zeros(self.K * self.M + self.K + _SLACK, **f32)
        self._v_flat = self._v_store[: self.K * self.M + self.K]
        self._e_store = None
        self._e_flat = None
        self._f32 = f32
        self.last_collective = None

    # ----- views
    @property
    def vertex_sum(self) -> torch.Tensor:
        return self._v_flat[: self.K * self.M].view(self.K, self.M)

    @property
    def vertex_count(self) -> torch.Tensor:
        return self._v_flat[self.K * self.M:]

    def _edges(self) -> torch.Tensor:
        if self._e_flat is None:
            n = self.K * self.n_max * self.n_max + self.K
            self._e_store = torch.zeros(n + _SLACK, **self._f32)
            self._e_flat = self._e_store[:n]
        return self._e_flat

    @property
    def edge_sum(self) -> torch.Tensor:
        return self._edges()[: self.K * self.n_max * self.n_max].view(self.K, self.n_max, self.n_max)

    @property
    def edge_count(self) -> torch.Tensor:
        return self._edges()[self.K * self.n_max * self.n_max:]

    # ----- accumulation (per batch, on this rank's images)
    def _accumulate(self, feat: torch.Tensor, label: torch.Tensor, sums: torch.Tensor, count: torch.Tensor):
        if not (feat.is_cuda and sums.is_cuda):
            raise RuntimeError("SchemaStatistics accumulates on the GPU only (no CPU fallback)")
        from cpp_extension import ops
        ops.stats_accumulate(feat, label.to(feat.device), sums, count)

    def add_vertices(self, full_vertices: torch.Tensor, label: torch.Tensor):
        """full_vertices [bs, M] = SchemaNet.feat_to_full_vertices(...) (init_schema_net.py:55-61)."""
        self._accumulate(full_vertices, label, self.vertex_sum, self.vertex_count)

    def add_edges(self, limited_edges: torch.Tensor, label: torch.Tensor):
        """limited_edges [bs, n_max, n_max] = SchemaNet.feat_to_limited_edges(...) (:31-35)."""
        self._accumulate(limited_edges, label, self.edge_sum, self.edge_count)

    # ----- cross-rank merge
    @staticmethod
    def collective_length(n: int, world: int) -> int:
        """length the reduce_scatter + all_gather form runs on: n rounded up to a multiple of the world size (the
        elements behind n are the zero slack of the flat buffer)"""
        return (n + world - 1) // world * world

    def _all_reduce_flat(self, store: torch.Tensor, n: int, large: bool):
        """SUM over ranks of store[:n].  `store` has >= _SLACK zero elements behind n."""
        rank, world = _world()
        if world == 1 and not (self.merge_single_rank and dist.is_available() and dist.is_initialized()):
            self.last_collective = None
            return
        if large and world <= _SLACK:
            # direct 1-hop exchange of 1/W shards on the xGMI mesh; the zero slack rounds the length up to the world size
            padded = store[: self.collective_length(n, world)]
            shard = torch.empty(padded.numel() // world, dtype=store.dtype, device=store.device)
            try:
                dist.reduce_scatter_tensor(shard, padded, op=dist.ReduceOp.SUM)
            except (RuntimeError, NotImplementedError, AttributeError) as exc:
                # a backend / torch version without reduce_scatter_tensor (older gloo, mpi) refuses before anything is
                # exchanged (every rank takes the same branch: the refusal does not depend on the data)
                logging.getLogger("SchemaStatistics").warning("reduce_scatter_tensor unavailable (%r): all_reduce instead", exc)
                dist.all_reduce(store[:n], op=dist.ReduceOp.SUM)
                self.last_collective = "all_reduce (reduce_scatter unavailable)"
            else:
                dist.all_gather_into_tensor(padded, shard)
                self.last_collective = "reduce_scatter+all_gather"
        else:
            dist.all_reduce(store[:n], op=dist.ReduceOp.SUM)
            self.last_collective = "all_reduce"
        if rank == 0:
            logging.getLogger("SchemaStatistics").debug("merged %d floats over %d ranks by %s", n, world, self.last_collective)

    def all_reduce_vertices(self):
        self._all_reduce_flat(self._v_store, self._v_flat.numel(), large=self._v_flat.numel() * 4 >= self.large_bytes)

    def all_reduce_edges(self):
        flat = self._edges()
        self._all_reduce_flat(self._e_store, flat.numel(), large=flat.numel() * 4 >= self.large_bytes)

    # ----- finalisation (identical on every rank after the all-reduce)
    def class_vertices(self) -> torch.Tensor:
        """init_schema_net.py:63-64: mean over the class's images, then row-normalise."""
        cv = self.vertex_sum / self.vertex_count[:, None]
        return cv / cv.sum(dim=-1, keepdim=True)

    def top_vertices(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """init_schema_net.py:116: (init_weights [K, n_max], valid_vertices i64 [K, n_max])."""
        return self.class_vertices().topk(self.n_max, dim=1)

    def class_edges(self) -> torch.Tensor:
        """init_schema_net.py:37-38: mean over the class's images (normalize() is the caller's)."""
        return self.edge_sum / self.edge_count[:, None, None]


@torch.no_grad()
def init_atlas(schema_net, batches, all_reduce: bool = True):
    """The reference's main() (:105-124) on an iterable of (ingredients, attn, attn_cls, label)
    batches that the caller has ALREADY sharded (see shard_indices).  `batches` is iterated
    twice.  Leaves schema_net initialised exactly like the reference script."""
    stats = SchemaStatistics(schema_net.num_classes, schema_net.num_vertices, schema_net.class_max_vertices,
                             device=schema_net.vertex_weights.tensor.device)
    for ing, attn, attn_cls, label in batches:
        stats.add_vertices(schema_net.feat_to_full_vertices(ing, attn_cls), label)
    if all_reduce:
        stats.all_reduce_vertices()
    init_weights, valid = stats.top_vertices()
    schema_net.register_class_vertices(valid)
    schema_net.vertex_weights.copy_(init_weights)
    for ing, attn, attn_cls, label in batches:
        stats.add_edges(schema_net.feat_to_limited_edges(ing, attn, label), label)
    if all_reduce:
        stats.all_reduce_edges()
    schema_net.edge_weights.copy_(stats.class_edges())
    schema_net.normalize()
    return stats
