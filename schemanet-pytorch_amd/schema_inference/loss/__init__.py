"""Drop-in for the reference's `schema_inference.loss` package (reference schema_inference/loss/__init__.py:3-16,
base_loss.py:9-34, schema_inference_loss.py:10-67): `Loss`, `CELoss`, `SchemaInferenceLoss`, `get_loss_fn` with the
same constructor arguments, the same keys in the returned dictionaries and the same registry names
("ce_loss", "schema_inference_loss").

`SchemaInferenceLoss` = cross entropy + the two sparsity terms of the atlas,
    entropy_vertex = max_k  H(class_vertices[k, :])          H(p) = -sum(p * log(p + 1e-7))
    entropy_edge   = mean_k max_i H(class_edges[k, i, :])
each also "rectified": x if x > a else a - 1 + 1 / (1 + a - x).  On the GPU the row entropies (a 105 MB operand at
K = 100, n = 512) come from `cpp_extension.ops.row_entropy` (one HIP pass forward, a recomputing backward that
skips the rows the max did not select); CPU tensors use the plain torch expression (the loss itself is not part of
the inference hot path and the reference trains on the device)."""
import os
from collections import OrderedDict
from pkgutil import extend_path
from typing import Any, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["Loss", "CELoss", "SchemaInferenceLoss", "get_loss_fn", "entropy", "rectify_linear"]
__path__ = extend_path(__path__, __name__)      # `schema_inference.loss.<submodule>` of a reference checkout behind us still resolves


def _logits_of(output: Dict[str, Any]) -> torch.Tensor:
    pred = output["pred"]
    return pred["pred"] if isinstance(pred, dict) else pred


def entropy(p: torch.Tensor, eps: float = 1.0e-7, dim: int = -1, keepdim: bool = False) -> torch.Tensor:
    if p.is_cuda and p.dtype == torch.float32 and dim in (-1, p.dim() - 1):
        from cpp_extension import ops
        pre = getattr(p, "_sn_row_entropy", None)      # class_edges of a training forward: computed by the pass that wrote them
        if pre is not None and pre[0] == eps and tuple(pre[1].shape) == tuple(p.shape[:-1]):
            return pre[1].unsqueeze(-1) if keepdim else pre[1]
        ent = ops.row_entropy(p, eps)
        return ent.unsqueeze(-1) if keepdim else ent
    return -(p * torch.log(p + eps)).sum(dim=dim, keepdim=keepdim)


def rectify_linear(x: torch.Tensor, a: float = 0) -> torch.Tensor:
    """x if x > a else a - 1 + 1 / (1 + a - x)   (reference schema_inference_loss.py:61-67: a python branch on a device scalar = a
    host synchronisation per term).  On the GPU the same value and gradient as a select, so that a training iteration stays
    asynchronous and can be captured into a hipGraph (train.GraphedTrainIter); the unselected branch's denominator is
    replaced by 1 before the division (at x = 1 + a it would be 0 and its - masked - gradient 0 * inf)."""
    if torch.is_tensor(x) and x.is_cuda:
        if x.dtype == torch.float32 and x.numel() < (1 << 30) and os.environ.get("SN_RECTIFY_FUSED", "1") != "0":
            from cpp_extension import ops
            return ops.rectify_linear(x, a)          # (round 6: one launch forward, one multiply back; the torch form below: seven and five)
        above = x > a
        return torch.where(above, x, a - 1 + 1.0 / torch.where(above, torch.ones_like(x), 1 + a - x))
    return x if x > a else a - 1 + 1.0 / (1 + a - x)


class Loss(nn.Module):
    def forward(self, output: Dict[str, torch.Tensor], target: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        raise NotImplementedError


class CELoss(Loss):
    def __init__(self, ignore_index: int = -100, reduction: str = "mean", **kwargs):
        super().__init__()
        self.loss_fn = nn.CrossEntropyLoss(ignore_index=ignore_index, reduction=reduction)

    def forward(self, output, target, name: str = "cls"):
        return OrderedDict([(name, self.loss_fn(_logits_of(output), target["label"]))])


class SchemaInferenceLoss(Loss):
    def __init__(self, re_a_vertex: float = 3, re_a_edge: float = 3, **kwargs):
        super().__init__()
        self.re_a_vertex, self.re_a_edge = re_a_vertex, re_a_edge

    def loss_sparsity(self, vertex_weights: torch.Tensor, edge_weights: torch.Tensor) -> Dict[str, torch.Tensor]:
        ev = entropy(vertex_weights).max(dim=0)[0]
        ee = entropy(edge_weights).max(dim=1)[0].mean()
        return OrderedDict([("entropy_vertex", ev), ("entropy_edge", ee),
                            ("re_entropy_vertex", rectify_linear(ev, a=self.re_a_vertex)),
                            ("re_entropy_edge", rectify_linear(ee, a=self.re_a_edge))])

    def forward(self, output, target):
        ret = OrderedDict(cls=F.cross_entropy(_logits_of(output), target["label"]))
        ret.update(self.loss_sparsity(output["class_vertices"], output["class_edges"]))
        return ret


_REGISTRY = {"ce_loss": CELoss, "schema_inference_loss": SchemaInferenceLoss}


def get_loss_fn(loss_cfg: Dict[str, Any], **kwargs) -> Loss:
    return _REGISTRY[loss_cfg["name"]](**loss_cfg.get("loss_cfg", dict()), **kwargs)
