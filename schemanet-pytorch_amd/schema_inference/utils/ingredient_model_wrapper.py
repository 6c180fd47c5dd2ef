"""Backbone taps -> inputs of the schema-inference path
(reference schema_inference/utils/ingredient_model_wrapper.py).
"""
import collections
from typing import Dict

import torch
import torch.nn as nn

from cpp_extension import ops


class IngredientModelWrapper(nn.Module):
    """Always in eval mode.  `backbone_jit(x)` must return {"mid_feat": [L+1, bs, D],
    "extracted": [bs*H, L+1, L+1] raw attention logits} (reference :45-47) and
    `discretization_jit` must behave like `discretization.DiscretizationModule`
    (reference :30, 48).

    forward(x): the reference's dict -- cls_token [bs,1,D], feat / feat_origin [bs,L,D],
      ingredients [bs,L], attn [bs,L,L], attn_cls [bs,L], all contiguous.
    taps(x): the subset the graph stage needs, without materialising anything: ingredients are
      written batch-first directly by the assignment kernel and attn / attn_cls are strided
      VIEWS of the backbone's [bs, H, L+1, L+1] tap (the head mean happens inside
      sn_instance_graph).
    """

    def __init__(self, backbone_jit: nn.Module, discretization_jit: nn.Module = None):
        super().__init__()
        self.backbone_jit = backbone_jit
        self.discretization_jit = discretization_jit
        self.register_buffer("discretization_tensor", discretization_jit.discretization.vocabulary.weight)
        self.num_ingredients: int = self.discretization_tensor.shape[0]
        self.emb_dim: int = self.discretization_tensor.shape[1]

    def train(self, mode: bool = True):
        self.training = mode
        for module in self.children():
            module.train(False)
        return self

    def eval(self):
        return self.train(False)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        ret: Dict[str, torch.Tensor] = collections.OrderedDict()
        out_backbone = self.backbone_jit(x)
        mid_feat = out_backbone["mid_feat"]
        extracted = out_backbone.get("extracted") if isinstance(out_backbone, dict) else None
        feat, ingredients = self.discretization_jit(mid_feat)
        ret["cls_token"] = feat[:1].transpose(0, 1)
        ret["feat"] = feat[1:].transpose(0, 1)
        ret["feat_origin"] = mid_feat[1:].transpose(0, 1)
        ret["ingredients"] = ingredients.transpose(0, 1)
        bs, L = ret["ingredients"].shape
        if extracted is not None:
            ret["attn"], ret["attn_cls"] = ops.head_mean_attention(extracted, bs)   # reference :58-68
        else:
            ret["attn"] = torch.zeros(bs, L, L, device=x.device)
            ret["attn_cls"] = torch.zeros(bs, L, device=x.device)
        for k, v in ret.items():
            ret[k] = v.contiguous()
        return ret

    @torch.no_grad()
    def taps(self, x: torch.Tensor, defer: bool = False) -> Dict[str, torch.Tensor]:
        return self.taps_from(self.backbone_jit(x), defer=defer)

    @torch.no_grad()
    def taps_from(self, out_backbone: Dict[str, torch.Tensor], defer: bool = False) -> Dict[str, torch.Tensor]:
        """The part of `taps` behind the backbone (no host synchronisation: SchemaNetPredictor captures it).
        defer=True (SchemaNetPredictor): `ingredients` holds the words of the fp16 screen and the dict carries `rerank`,
        the handle with which `SchemaNet.instance_graph_padded` finishes the undecided ones inside its kernel; the caller
        MUST pass it on (or call `rerank.finish()`).  Default: the ids are final on return."""
        mid_feat = out_backbone["mid_feat"]                     # [L+1, bs, D] sequence-first
        extracted = out_backbone["extracted"]                   # [bs*H, L+1, L+1]
        Lp1, bs, _ = mid_feat.shape
        L = Lp1 - 1
        ingredients = torch.empty((bs, L), dtype=torch.int64, device=mid_feat.device)
        # tokens [L, bs, D] view; word ids land transposed ([bs, L]) through the output strides
        rerank = None
        if defer:
            _, rerank = self.discretization_jit.discretization.assign(mid_feat[1:], out=ingredients.t(), defer=True)
        else:
            self.discretization_jit.discretization.assign(mid_feat[1:], out=ingredients.t())
        heads = extracted.reshape(bs, -1, Lp1, Lp1)
        ret = {"ingredients": ingredients, "attn": heads[:, :, 1:, 1:], "attn_cls": heads[:, :, 0, 1:]}
        if rerank is not None:
            ret["rerank"] = rerank
        return ret
