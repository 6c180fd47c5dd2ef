"""Visual-word discretization on the MI355X matrix cores.

`Discretization.encode` of the reference (discretization/discretization.py:58-70) materialises
the [n*bs, size] distance matrix with torch.cdist and takes argmin.  Here the assignment is one
fused HIP launch pair (fp16-MFMA screen + fp64 re-rank, csrc/sn_assign.hip) that reads the
tokens once and writes only the int64 word ids; the distance matrix never exists in memory.
"""
import logging
from typing import Tuple

import torch
import torch.nn as nn

from cpp_extension import ops

from .adapter import Adapter


class Discretization(nn.Module):
    """Same constructor, attributes (`vocabulary`, `size`, `dim`), state-dict key
    (`vocabulary.weight`) and methods as the reference class."""

    def __init__(
        self,
        size: int,
        dim: int,
        detach_input_seq: bool = True,
        uniform_range: Tuple[float, float] = (-1, 1),
        exact: bool = False,
    ):
        super().__init__()
        self.logger = logging.getLogger("discretization")
        self.size = size
        self.dim = dim
        self.detach_input_seq = detach_input_seq
        self.exact = exact            # True: fp64 full scan instead of MFMA screen + re-rank
        self.vocabulary = nn.Embedding(size, dim)
        nn.init.uniform_(self.vocabulary.weight, uniform_range[0], uniform_range[1])
        self._packed = ops.PackedCodebook()
        self.activate()

    def initial_vocabulary(self, vocabulary_fp: str):
        """Load a k-means codebook file `cluster_{M}_from_{N}.pth` (reference :40-48)."""
        vocabulary: torch.Tensor = torch.load(vocabulary_fp, map_location="cpu")
        if vocabulary.shape[0] > self.size:
            self.logger.warning("Too much external vocabulary, using random picked vocabulary...")
            vocabulary = vocabulary[torch.randperm(vocabulary.shape[0])][:self.size]
        with torch.no_grad():
            self.vocabulary.weight.copy_(vocabulary)
        self.invalidate()

    def invalidate(self):
        """Forget the packed (fp16 fragment) image of the codebook: the next `assign` re-packs it.  Needed only after
        writes that do not bump the tensor's version counter (`vocabulary.weight.data.copy_(...)`)."""
        self._packed.invalidate()

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        ret = super().load_state_dict(state_dict, strict, **kw)
        self.invalidate()
        return ret

    def _load_from_state_dict(self, *args, **kw):       # also when loaded as a sub-module of a larger state dict
        super()._load_from_state_dict(*args, **kw)
        self._packed.invalidate()

    def deactivate(self):
        self._activate = False

    def activate(self):
        self._activate = True

    def assign(self, seq: torch.Tensor, out: torch.Tensor = None, defer: bool = False):
        """seq [n_outer, n_inner, dim] (any outer strides) -> word ids int64 [n_outer, n_inner].
        defer=True: -> (ids, handle): the tokens the fp16 screen could not decide are left to the consumer of the ids
        (`SchemaNet.instance_graph_padded(..., rerank=handle)` finishes them inside its kernel, `handle.finish()` by the
        stand-alone re-rank); handle is None where everything is final already."""
        codebook, packed = self._packed.get(self.vocabulary.weight)
        return ops.assign_words(seq, codebook, packed, out=out, mode=1 if self.exact else 0, defer=defer)

    def encode(self, seq: torch.Tensor) -> Tuple[torch.Tensor, torch.LongTensor]:
        if self.detach_input_seq:
            seq = seq.detach()
        n, bs = seq.shape[:2]
        ingredients = self.assign(seq.reshape(n, bs, self.dim) if seq.dim() != 3 else seq)
        if self._activate:
            seq = self.vocabulary(ingredients)
        return seq.reshape(n, bs, self.dim), ingredients

    def forward(self, seq: torch.Tensor) -> Tuple[torch.Tensor, torch.LongTensor]:
        """seq [n, bs, dim] -> (encoded seq [n, bs, dim], word ids [n, bs])."""
        assert int(seq.shape[2]) == self.dim, f"dimension {seq.shape[2]} not match to {self.dim}"
        return self.encode(seq)


class DiscretizationModule(nn.Module):
    """Stand-in for the TorchScript file `discretization-jit.pth`
    (reference scripts/save_backbone_jit.py:121-131, traced at :195-198): called as
    `module(mid_feat[L+1, bs, D]) -> (feat[L+1, bs, D], ingredients[L, bs])` and read for
    `module.discretization.vocabulary.weight` (ingredient_model_wrapper.py:30, 48)."""

    def __init__(self, discretization: Discretization):
        super().__init__()
        self.discretization = discretization

    def forward(self, mid_feat: torch.Tensor):
        adapter = Adapter()
        seq = adapter.adapt(mid_feat)
        output, match = self.discretization(seq)
        return adapter.reconstruct(output, match)

    @torch.no_grad()
    def assign_batch_first(self, tokens: torch.Tensor) -> torch.LongTensor:
        """tokens [B, L+1, D] batch-first (cls at index 0) -> word ids [B, L] without the
        transposes / cat of the sequence-first contract (what SchemaNetPredictor uses)."""
        return self.discretization.assign(tokens[:, 1:, :])
