"""Small tensor helpers of the IR-graph code (reference schema_inference/graph/utils.py)."""
import os
from typing import Iterable

import torch
import torch.nn as nn


def _safe_div(x: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    return (x / d).nan_to_num(0)


def normalize_sum_(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """In place: x <- x / x.sum(dim), NaN -> 0  (reference utils.py:7-13)."""
    x.div_(x.sum(dim=dim, keepdim=True))
    return x.nan_to_num_(0)


def normalize_max_(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """In place: x <- x / x.max(dim), NaN -> 0  (reference utils.py:16-22)."""
    x.div_(x.max(dim=dim, keepdim=True)[0])
    return x.nan_to_num_(0)


def normalize_sum(x: torch.Tensor, dim: int = -1, detach_sum: bool = False) -> torch.Tensor:
    """Out of place; with detach_sum no gradient flows through the denominator (utils.py:25-34)."""
    total = x.sum(dim=dim, keepdim=True)
    return _safe_div(x, total.detach() if detach_sum else total)


def normalize_max(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    return _safe_div(x, x.max(dim=dim, keepdim=True)[0])


def normalize_sum_clamp(x: torch.Tensor, dim: int = -1, detach_sum: bool = False, min_val: float = 0) -> torch.Tensor:
    """clamp_min then normalize_sum (utils.py:46-52)."""
    return normalize_sum(x.clamp_min(min_val), dim, detach_sum=detach_sum)


def pair_wise_point_dist(h: int, w: int, pow: float = 2, device: torch.device = None) -> torch.Tensor:
    """[h*w, h*w] p-norm distances between the integer grid points of an h x w feature map,
    row-major flattening (utils.py:55-69).  Computed from coordinate differences, which for
    integer coordinates gives the same bits as the reference's torch.cdist."""
    rows = torch.arange(h, dtype=torch.float, device=device).repeat_interleave(w)
    cols = torch.arange(w, dtype=torch.float, device=device).repeat(h)
    dr = (rows[:, None] - rows[None, :]).abs()
    dc = (cols[:, None] - cols[None, :]).abs()
    if pow == 2:
        return (dr * dr + dc * dc).sqrt()
    return (dr.pow(pow) + dc.pow(pow)).pow(1.0 / pow)


def pair_wise_point_sim(h: int, w: int, alpha: float = 1, pow: float = 2, device: torch.device = None) -> torch.Tensor:
    """sim = 1 / (1 + dist / alpha)  (utils.py:72-81)."""
    assert alpha >= 0
    return 1 / (1 + pair_wise_point_dist(h, w, pow, device) / alpha)


class MyParameter(nn.Module):
    """Holder whose attribute name `tensor` fixes the state-dict key `<name>.tensor`
    (utils.py:84-106).  as_buffer=True stores a non-trainable Parameter."""

    def __init__(self, shape: Iterable[int], dtype=torch.float, as_buffer: bool = False) -> None:
        super().__init__()
        self.tensor = nn.Parameter(torch.zeros(tuple(shape), dtype=dtype), requires_grad=not as_buffer)

    def copy_(self, value: torch.Tensor):
        with torch.no_grad():
            self.tensor.copy_(value)

    def normalize_sum_(self, dim: int, min_val: float = 0, zero_diagonal: bool = False):
        """reference utils.py:96-99; zero_diagonal: the `diagonal().fill_(0)` that schema_net.py:141-142 applies next, folded in."""
        t = self.tensor
        with torch.no_grad():
            if (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and dim in (-1, t.dim() - 1) and t.numel() >= (1 << 20)
                    and os.environ.get("SN_NORMALIZE_FUSED", "1") != "0"):
                # a large parameter (edge_weights: 404 MB at config [4]'s real size): one pass instead of four (csrc/sn_train.hip)
                from cpp_extension import ops
                ops.normalize_sum_rows_(t, min_val, zero_diagonal=zero_diagonal)
                torch.autograd.graph.increment_version(t)        # (what an in-place torch op would have done: caches key on it)
                return
            normalize_sum_(t.clamp_min_(min_val), dim=dim)
            if zero_diagonal:
                t.diagonal(dim1=1, dim2=2).fill_(0)
