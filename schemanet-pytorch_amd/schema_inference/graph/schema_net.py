"""IR-Atlas parameters and instance IR-graph construction, HIP-backed.

Same class name, constructor arguments, attributes and state-dict keys as the reference
(`schema_inference/graph/schema_net.py`), so reference checkpoints / init-atlas files load
unchanged and `worker_schema_net.py:343-364`, `scripts/init_schema_net.py:100-120` can build it
the same way.  What changed is where the work happens: the reference round-trips every batch
GPU -> CPU -> C++ -> GPU three times (its schema_net.py:314-315, 367-369); here one
`sn_instance_graph` launch per batch turns attention *logits* into the padded instance graph
without leaving HBM.
"""
import logging
from typing import Dict, List, Optional, Tuple

import os

import torch
import torch.nn as nn

from cpp_extension import ops

from . import utils as graph_utils


class InstanceLists(dict):
    """The reference's dictionary of three per-image lists (schema_net.py:377-399) that also remembers the padded batch its
    entries are views of (`padded`, or None).  `Matcher.forward` pads the lists to the batch maximum like the reference; when
    every entry still is the view this class made, the padded batch IS that result and 3 x bs pads, their zero fills and -
    in training - 3 x bs slice gradients (each a zero fill of the whole batch plus an add) are not launched."""
    padded = None


class SchemaNet(nn.Module):
    """
    Parameters (state-dict keys `<name>.tensor`):
        class_ingredients [K, n_max] i64, vertex_weights [K, n_max], edge_weights [K, n_max, n_max],
        vertex_attribute_weights [2, 1], edge_attribute_weights [2, 1]
    Extra (not in the reference): `n_pad` policy for the padded instance graphs, see
    `instance_graph_padded`.
    """

    def __init__(
        self,
        num_vertices: int,
        num_classes: int = 10,
        dist_alpha: float = 1,
        dist_pow: float = 2,
        feat_h: int = 14,
        feat_w: int = 14,
        class_max_vertices: int = None,
        constant_vertex_attr: Tuple[float, float] = None,
        constant_edge_attr: Tuple[float, float] = None,
        clamp_vertex_attn: float = None,
        clamp_edge_attn: float = None,
        remove_self_loop: bool = False,
        prune_node_threshold: float = None,
        apply_normalize: bool = True,
        clamp_weights: bool = True,
    ):
        super().__init__()
        self.logger = logging.getLogger("SchemaNet")
        self.num_vertices = num_vertices
        self.num_classes = num_classes
        self.dist_alpha = dist_alpha
        self.dist_pow = dist_pow
        self.feat_h = feat_h
        self.feat_w = feat_w
        self.constant_vertex_attr = constant_vertex_attr
        self.constant_edge_attr = constant_edge_attr
        self.clamp_vertex_attn = clamp_vertex_attn
        self.clamp_edge_attn = clamp_edge_attn
        self.remove_self_loop = remove_self_loop
        self.prune_node_threshold = prune_node_threshold
        self.apply_normalize = apply_normalize
        self.clamp_weights = clamp_weights

        self.register_buffer("n_tracked", torch.zeros(num_classes), persistent=False)
        if class_max_vertices is None:
            class_max_vertices = num_vertices
        assert class_max_vertices <= num_vertices
        self.class_max_vertices = class_max_vertices

        P = graph_utils.MyParameter
        self.class_ingredients = P((num_classes, class_max_vertices), dtype=torch.long, as_buffer=True)
        self.vertex_weights = P((num_classes, class_max_vertices))
        self.edge_weights = P((num_classes, class_max_vertices, class_max_vertices))
        self.vertex_attribute_weights = P((2, 1), as_buffer=constant_vertex_attr is not None)
        self.edge_attribute_weights = P((2, 1), as_buffer=constant_edge_attr is not None)
        # dense word -> slot table per class (device side form of class_ingredient_dict)
        self.register_buffer("class_slot", torch.full((num_classes, num_vertices), -1, dtype=torch.int32),
                             persistent=False)
        self._class_dict_cache: Optional[List[Dict[int, int]]] = None
        self._registered = False
        self._reset_parameters()

    # ------------------------------------------------------------------ parameters
    def _reset_parameters(self):
        """reference schema_net.py:104-119"""
        nn.init.constant_(self.vertex_attribute_weights.tensor, 0.5)
        nn.init.constant_(self.edge_attribute_weights.tensor, 0.5)
        nn.init.trunc_normal_(self.vertex_weights.tensor, mean=0.5, std=1 / 6, a=0, b=1)
        nn.init.trunc_normal_(self.edge_weights.tensor, mean=0.5, std=1 / 6, a=0, b=1)
        self.vertex_weights.normalize_sum_(dim=-1)
        self.edge_weights.normalize_sum_(dim=-1)
        if self.constant_vertex_attr is not None:
            self.vertex_attribute_weights.copy_(torch.tensor(self.constant_vertex_attr).reshape(2, 1))
        if self.constant_edge_attr is not None:
            self.edge_attribute_weights.copy_(torch.tensor(self.constant_edge_attr).reshape(2, 1))
        self.normalize()

    def register_class_vertices(self, class_vertices: torch.LongTensor):
        """Fix which words make up each class graph (reference :121-126).  Builds the dense
        int32 [K, M] slot table with one scatter instead of K python dicts."""
        self.class_ingredients.copy_(class_vertices)
        ci = self.class_ingredients.tensor
        slot = torch.full_like(self.class_slot, -1)
        src = torch.arange(ci.shape[1], dtype=torch.int32, device=ci.device).expand_as(ci)
        valid = (ci >= 0) & (ci < self.num_vertices)
        slot.scatter_(1, ci.clamp(0, self.num_vertices - 1), torch.where(valid, src, torch.full_like(src, -1)))
        self.class_slot.copy_(slot)
        self._class_dict_cache = None
        self._registered = True

    @property
    def class_ingredient_dict(self) -> List[Dict[int, int]]:
        """K dicts {word: slot}, built on demand (the reference attribute, schema_net.py:82)."""
        if not self._registered:
            return []
        if self._class_dict_cache is None:
            self._class_dict_cache = [
                {int(k): v for v, k in enumerate(row)} for row in self.class_ingredients.tensor.tolist()]
        return self._class_dict_cache

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        ret = super().load_state_dict(state_dict, strict)
        self.register_class_vertices(self.class_ingredients.tensor)
        return ret

    @torch.no_grad()
    def normalize(self):
        """Called before every training iteration (worker_schema_net.py:127; reference :133-142)."""
        if self.clamp_weights:
            self.vertex_attribute_weights.tensor.clamp_(min=0.01, max=10)
            self.edge_attribute_weights.tensor.clamp_(min=0.01, max=10)
        if self.apply_normalize:
            self.vertex_weights.normalize_sum_(dim=-1)
            self.edge_weights.normalize_sum_(dim=-1, zero_diagonal=self.remove_self_loop)

    # ------------------------------------------------------------------ atlas
    def _needs_grad(self, *tensors) -> bool:
        return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)

    def get_class_vertices(self, detach: bool = False) -> torch.Tensor:
        vw = self.vertex_weights.tensor.detach() if detach else self.vertex_weights.tensor
        return graph_utils.normalize_sum_clamp(vw, detach_sum=True, min_val=1.0e-5)

    def get_class_edges(self, detach: bool = False) -> torch.Tensor:
        """Differentiable form (reference :152-175) used when gradients are required."""
        ew = self.edge_weights.tensor.detach() if detach else self.edge_weights.tensor
        if ew.is_cuda and ew.is_contiguous() and self._needs_grad(ew) and os.environ.get("SN_ATLAS_AUTOGRAD_FUSED", "1") != "0":
            # one HIP pass forward (pruning the parameter in place, :164) and one back, the gradients - NaN rows included -
            # those of the chain of torch ops below
            # (training: the row entropies the loss's sparsity term wants ride along, SN_ATLAS_ENTROPY_FUSED=0 turns that off)
            return ops.class_edges_autograd(ew, self.vertex_weights.tensor.detach(), self.prune_node_threshold, self.remove_self_loop,
                                            with_entropy=os.environ.get("SN_ATLAS_ENTROPY_FUSED", "1") != "0")
        if self.prune_node_threshold is not None:
            with torch.no_grad():
                keep = self.get_class_vertices(detach=True) > self.prune_node_threshold
                mask = keep[:, :, None] & keep[:, None, :]
                ew.masked_fill_(~mask, 0)                      # in place on the Parameter (:164)
            ew = ew * mask.to(ew.dtype)
        ew = graph_utils.normalize_sum_clamp(ew, detach_sum=True)
        if self.remove_self_loop:
            eye = torch.eye(ew.shape[-1], dtype=torch.bool, device=ew.device)
            ew = ew.masked_fill(eye, 0)
        return ew

    def get_atlas(self, detach: bool = False, fused_adjacency: bool = False) -> Dict[str, torch.Tensor]:
        """reference :177-184.  Without autograd the whole normalisation is one fused HIP pass
        over edge_weights (csrc/sn_atlas.hip), including the in-place pruning.

        fused_adjacency=True (inference only) skips the [K, n, n] `class_edges` tensor: the dict
        then carries `class_adjacency` = the GCN operand (E + E^T)/2 + I as split-fp16 planes, built
        straight from the pruned parameters (same values, one pass less over the atlas); `Matcher`
        consumes it directly.  fused_adjacency="compact": the same for a pruned atlas with the operand compacted to the kept
        vertices of each class (ops.atlas_adjacency_planes_compact; falls back to True where it does not apply).
        fused_adjacency="with_edges": both - `class_edges` is written by the same pass (within
        one rounding of the unfused route: w * (1 / row sum) instead of w / row sum); what SchemaNetPredictor uses."""
        vw, ew = self.vertex_weights.tensor, self.edge_weights.tensor
        if fused_adjacency and vw.is_cuda and (detach or not self._needs_grad(vw, ew)):
            if (fused_adjacency == "compact" and self.prune_node_threshold is not None and vw.shape[1] <= 1024
                    and os.environ.get("SN_ATLAS_COMPACT", "1") != "0" and self._atlas_compaction_pays()):
                # a pruned atlas, compacted: the operand holds the kept vertices of every class only (class_perm / class_n_kept
                # say which); `Matcher` adds the isolated vertices' share of the class feature without a product
                # (the in-place pruning is idempotent: once it has run on these versions of the two parameters the rows of the
                # pruned vertices are zero and are not read again - a weight-only fact like the packed codebook of S1)
                pkey = (vw.data_ptr(), vw._version, ew.data_ptr(), ew._version, self.prune_node_threshold)
                cv, adj, perm, n_kept = ops.atlas_adjacency_planes_compact(vw.detach(), ew.detach(), self.prune_node_threshold,
                                                                           self.remove_self_loop,
                                                                           pruned_rows_are_zero=getattr(self, "_pruned_in_place", None) == pkey)
                self._pruned_in_place = pkey
                return {"class_vertices": cv, "class_adjacency": adj, "class_ingredients": self.class_ingredients.tensor,
                        "class_perm": perm, "class_n_kept": n_kept}
            if fused_adjacency == "with_edges":      # the reference's dictionary AND the GCN operand from one pass over the atlas
                cv, adj, ce = ops.atlas_adjacency_planes(vw.detach(), ew.detach(), self.prune_node_threshold, self.remove_self_loop,
                                                         want_edges=True)
                return {"class_vertices": cv, "class_edges": ce, "class_ingredients": self.class_ingredients.tensor,
                        "class_adjacency": adj}
            cv, adj = ops.atlas_adjacency_planes(vw.detach(), ew.detach(), self.prune_node_threshold, self.remove_self_loop)
            return {"class_vertices": cv, "class_adjacency": adj, "class_ingredients": self.class_ingredients.tensor}
        if vw.is_cuda and (detach or not self._needs_grad(vw, ew)):
            cv, ce = ops.atlas_normalize(vw.detach(), ew.detach(), self.prune_node_threshold, self.remove_self_loop)
        else:
            cv, ce = self.get_class_vertices(detach), self.get_class_edges(detach)
        out = {"class_vertices": cv, "class_edges": ce, "class_ingredients": self.class_ingredients.tensor}
        kcv = getattr(ce, "_sn_kernel_cv", None)
        # SN_TRAIN_COMPACT: 0 never; 1 (default) where the caller has asked for it (`compact_training`: train.GraphedTrainIter does -
        # the compacted route has ~90 launches more per iteration, which a captured iteration does not pay for and an eager one does:
        # 6.5 ms against 5.3 ms eager, 5.0 against 5.4 ms captured at config [4]'s real size) and it pays (below); 2 always (tests)
        mode = os.environ.get("SN_TRAIN_COMPACT", "1")
        if (kcv is not None and self.prune_node_threshold is not None and vw.shape[1] <= 1024 and mode != "0"
                and (mode == "2" or (getattr(self, "compact_training", False) and self._train_compaction_pays(kcv)))):
            # training with a pruned atlas (round 5): the partition of every class into kept vertices (first) and pruned ones, from the
            # normalised weights the pruning pass itself compared with the threshold; `Matcher` runs the class GNN on the kept vertices
            out["class_perm"], out["class_n_kept"] = ops.atlas_keep_perm(kcv, self.prune_node_threshold)
        return out

    def __getstate__(self):
        """copies and pickles of the module start the bookkeeping of `_train_compaction_pays` afresh"""
        state = self.__dict__.copy()
        state.pop("_compaction_state", None)
        return state

    def _train_compaction_pays(self, kernel_cv: torch.Tensor) -> bool:
        """Compacting the class graphs pays when enough vertices are pruned (the products cost row tiles x k stages per class; the
        compacted route adds a permuted operand, the scatter of the edge gradient and the per-word pass of the pruned vertices):
        below 3/4 of the vertices kept.  The kept fraction is read from the device at the first training forward and at every
        256th from then on (one host synchronisation each: the vertex weights change with every optimizer step, so a per-version
        look-up as in inference would synchronise every iteration); in between - and inside a stream capture - the last decision
        holds.  Deterministic in the number of calls: two runs from the same state take the same route."""
        st = getattr(self, "_compaction_state", None)
        if st is None:
            st = self._compaction_state = {"calls": 0, "decision": False}
        if torch.cuda.is_current_stream_capturing():
            return st["decision"]
        if st["calls"] % 256 == 0:
            with torch.no_grad():
                st["decision"] = bool(float((kernel_cv > float(self.prune_node_threshold)).float().mean()) < 0.75)
        st["calls"] += 1
        return st["decision"]

    def _atlas_compaction_pays(self) -> bool:
        """Are enough class vertices under prune_node_threshold for the compacted route to pay (see below)?  One host
        synchronisation per VERSION of vertex_weights - like the
        packed codebook of S1 a weight-only fact, looked up from then on; `p.data` writes: `invalidate_pruned_flag()`."""
        vw = self.vertex_weights.tensor
        key = (vw.data_ptr(), vw._version, vw.device)
        if getattr(self, "_pruned_flag", None) is None or self._pruned_flag[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return False                                    # (cannot ask the device now: the plain route is always right)
            with torch.no_grad():
                c = vw.detach().clamp_min(1.0e-5)
                c = (c / c.sum(dim=-1, keepdim=True)).nan_to_num(0)
                # ... and does compaction PAY?  The products cost row tiles (128) x k stages (32) per class; the compacted route
                # adds two bookkeeping launches and builds its operand through a permutation (4-byte gathers instead of
                # 16-byte rows: 64 us against 40 for the bench's class graphs).  A freshly initialised atlas of 512 vertices has
                # ~36 of them under the threshold: four row tiles either way - the plain route.  Taken below 3/4 of the work.
                n = vw.shape[1]
                n_kept = (c > float(self.prune_node_threshold)).sum(dim=1)
                work = lambda m: ((m + 127) // 128) * ((m + 31) // 32)       # noqa: E731
                kept_work = int(work(n_kept).sum().item())
                full_work = int(vw.shape[0]) * int(work(torch.tensor(n)))
                self._pruned_flag = (key, bool(kept_work * 4 <= full_work * 3))
        return self._pruned_flag[1]

    def invalidate_pruned_flag(self):
        self._pruned_flag = None
        self._pruned_in_place = None

    # ------------------------------------------------------------------ initialisation path
    def _dev(self) -> torch.device:
        d = self.vertex_weights.tensor.device
        if d.type != "cuda":
            raise RuntimeError("SchemaNet must live on the GPU (`.cuda()`); the HIP path has no CPU fallback")
        return d

    def feat_to_full_vertices(self, ingredients: torch.LongTensor, attn_cls: torch.Tensor) -> torch.Tensor:
        """[bs, L] words, [bs, L] cls-attention LOGITS -> vertex weights over all M words
        [bs, num_vertices] (reference :188-207: clamp, softmax WITHOUT nan_to_num, dense
        (count, mean) attributes, normalize_max_(dim=1), @ w_v)."""
        dev = self._dev()
        if self.clamp_vertex_attn is not None:
            attn_cls.masked_fill_(attn_cls < self.clamp_vertex_attn, float("-inf"))   # side effect kept
        w = self.vertex_attribute_weights.tensor
        ing = ingredients.to(dev)
        if self._needs_grad(w):
            attr2, _ = ops.full_vertices(ing, attn_cls.to(dev), self.num_vertices, is_logits=True, clamp=None,
                                         want_attr2=True, want_weighted=False)
            graph_utils.normalize_max_(attr2, dim=1)
            return ops.weigh_attributes(attr2, w)
        _, v = ops.full_vertices(ing, attn_cls.to(dev), self.num_vertices, w_v=w, is_logits=True, clamp=None)
        return v

    def feat_to_limited_edges(self, ingredients: torch.LongTensor, attn: torch.Tensor, label: torch.LongTensor) -> torch.Tensor:
        """[bs, L] words, [bs, L, L] attention LOGITS, [bs] labels -> edges in the slot layout of
        each image's class [bs, n_max, n_max] (reference :222-254)."""
        assert self._registered, "run `register_class_vertices` before"
        dev = self._dev()
        w = self.edge_attribute_weights.tensor
        kw = dict(is_logits=True, clamp=self.clamp_edge_attn, feat_h=self.feat_h, feat_w=self.feat_w,
                  dist_alpha=self.dist_alpha, dist_pow=self.dist_pow, remove_self_loop=self.remove_self_loop)
        ing, at, lab = ingredients.to(dev), attn.to(dev), label.to(dev)
        if self._needs_grad(w):
            attr2, _ = ops.limited_edges(ing, at, self.class_slot, lab, self.class_max_vertices,
                                         want_attr2=True, want_weighted=False, **kw)
            graph_utils.normalize_sum_(attr2, dim=2)
            if self.remove_self_loop:
                attr2.diagonal(dim1=1, dim2=2).fill_(0)
            return ops.weigh_attributes(attr2, w)
        _, e = ops.limited_edges(ing, at, self.class_slot, lab, self.class_max_vertices, w_e=w, **kw)
        return e

    # ------------------------------------------------------------------ prediction path
    def default_n_pad(self, L: int) -> int:
        return min(L, self.num_vertices)

    def instance_graph_padded(self, ingredients: torch.LongTensor, attn: torch.Tensor, attn_cls: torch.Tensor,
                              n_pad: int = None, mutate_inputs: bool = True, zero_padding: bool = True,
                              return_attn_cls: bool = False, rerank=None) -> Dict[str, torch.Tensor]:
        """One fused launch: logits -> padded instance graphs.

        ingredients [bs, L] i64; attn [bs, L, L] or [bs, H, L, L] logits (head mean fused);
        attn_cls [bs, L] or [bs, H, L] logits.  Any batch / head / row strides (e.g. slices of the
        backbone's [bs*H, L+1, L+1] tap) are consumed in place.
        Returns ids [bs, n_pad] (pad = num_vertices), vertices [bs, n_pad], edges [bs, n_pad, n_pad]
        (pads 0), n [bs] i32, n_max [1] i32 -- the layout Matcher builds with F.pad
        (reference match.py:48-54), without the .tolist() syncs.
        mutate_inputs: reproduce the reference's in-place masked_fill_ on `attn_cls`
        (schema_net.py:296) when it is a plain [bs, L] contiguous tensor.
        zero_padding=False: the rows / columns of `edges` beyond an image's own vertex count are left unwritten
        (about two thirds of the padded batch are such zeros) and the dict says `edges_padded: False`;
        `Matcher.forward_padded` masks by `n` instead.  For callers that only want the scores.
        rerank: the handle of `Discretization.assign(..., defer=True)` that produced `ingredients`: the word ids the fp16
        screen left undecided are finished inside this launch (and written back to `ingredients`).
        return_attn_cls: also return `attn_cls` [bs, L] = the head-averaged cls-attention logits after the clamp
        (< clamp_vertex_attn -> -inf), i.e. what the reference's `attn_cls` holds after schema_net.py:296, when the
        input is a per-head view that cannot be masked in place.
        """
        dev = self._dev()
        B, L = ingredients.shape
        n_pad = n_pad or self.default_n_pad(L)
        w_v, w_e = self.vertex_attribute_weights.tensor, self.edge_attribute_weights.tensor
        need_grad = self._needs_grad(w_v, w_e)
        masked_out = None
        if (mutate_inputs and self.clamp_vertex_attn is not None and attn_cls.dim() == 2
                and attn_cls.is_contiguous() and attn_cls.dtype == torch.float32 and attn_cls.device == dev):
            masked_out = attn_cls
        elif return_attn_cls:
            masked_out = torch.empty((B, L), dtype=torch.float32, device=dev)
        g = ops.instance_graph(
            ingredients.to(dev), attn.to(dev), attn_cls.to(dev), w_v=w_v, w_e=w_e, n_pad=n_pad,
            pad_id=self.num_vertices, attn_is_logits=True, attn_cls_is_logits=True,
            clamp_v=self.clamp_vertex_attn, clamp_e=self.clamp_edge_attn, feat_h=self.feat_h, feat_w=self.feat_w,
            dist_alpha=self.dist_alpha, dist_pow=self.dist_pow, mean=True, remove_self_loop=self.remove_self_loop,
            want_attr2=need_grad, want_weighted=not need_grad, attn_cls_masked_out=masked_out,
            zero_padding=zero_padding or need_grad, rerank=rerank)
        if need_grad:   # keep `@ w` visible to autograd (the reference does it inside C++)
            g["v"] = ops.weigh_attributes(g["v2"], w_v)
            g["e"] = ops.weigh_attributes(g["e2"], w_e)
        ret = {"ids": g["ids"], "vertices": g["v"], "edges": g["e"], "n": g["n"], "n_max": g["n_max"],
               "edges_padded": bool(zero_padding or need_grad)}
        if return_attn_cls:
            if masked_out is None:          # no clamp configured: the plain head mean
                masked_out = attn_cls.to(dev, torch.float32) if attn_cls.dim() == 2 else attn_cls.to(dev, torch.float32).mean(dim=1)
            ret["attn_cls"] = masked_out
        return ret

    @staticmethod
    def _as_lists(g: Dict[str, torch.Tensor]) -> Dict[str, List[torch.Tensor]]:
        sizes = g["n"].tolist()                                   # host sync, as in the reference (:302)
        out = InstanceLists({
            "instance_ingredients": [g["ids"][b, :n] for b, n in enumerate(sizes)],
            "instance_vertices": [g["vertices"][b, :n] for b, n in enumerate(sizes)],
            "instance_edges": [g["edges"][b, :n, :n] for b, n in enumerate(sizes)],
        })
        if g.get("edges_padded", False):                          # (the padding holds the pad id / zeros: Matcher may use the batch as it is)
            out.padded = {"ids": g["ids"], "vertices": g["vertices"], "edges": g["edges"], "n": g["n"], "sizes": sizes}
        return out

    def feat_to_instance_vertices(self, ingredients: torch.LongTensor, attn_cls: torch.Tensor
                                  ) -> Tuple[List[torch.LongTensor], List[torch.Tensor]]:
        """reference :278-305 (lists of per-image tensors)."""
        dev = self._dev()
        B, L = ingredients.shape
        w_v = self.vertex_attribute_weights.tensor
        need_grad = self._needs_grad(w_v)
        masked_out = attn_cls if (self.clamp_vertex_attn is not None and attn_cls.is_contiguous()
                                  and attn_cls.device == dev and attn_cls.dtype == torch.float32) else None
        g = ops.instance_graph(ingredients.to(dev), None, attn_cls.to(dev), w_v=w_v, n_pad=self.default_n_pad(L),
                               pad_id=self.num_vertices, attn_cls_is_logits=True, clamp_v=self.clamp_vertex_attn,
                               want_attr2=need_grad, want_weighted=not need_grad, attn_cls_masked_out=masked_out)
        v = ops.weigh_attributes(g["v2"], w_v) if need_grad else g["v"]
        sizes = g["n"].tolist()
        return ([g["ids"][b, :n] for b, n in enumerate(sizes)], [v[b, :n] for b, n in enumerate(sizes)])

    def feat_to_instance_edges(self, ingredients: torch.LongTensor, attn: torch.Tensor,
                               instance_ingredients: List[torch.LongTensor] = None) -> List[torch.Tensor]:
        """reference :320-356.  `instance_ingredients` is accepted for signature parity; the row
        order it induces (rank among the image's sorted words) is what the kernel produces."""
        dev = self._dev()
        B, L = ingredients.shape
        w_e = self.edge_attribute_weights.tensor
        need_grad = self._needs_grad(w_e)
        g = ops.instance_graph(ingredients.to(dev), attn.to(dev), None, w_e=w_e, n_pad=self.default_n_pad(L),
                               pad_id=self.num_vertices, attn_is_logits=True, clamp_e=self.clamp_edge_attn,
                               feat_h=self.feat_h, feat_w=self.feat_w, dist_alpha=self.dist_alpha,
                               dist_pow=self.dist_pow, remove_self_loop=self.remove_self_loop,
                               want_attr2=need_grad, want_weighted=not need_grad)
        e = ops.weigh_attributes(g["e2"], w_e) if need_grad else g["e"]
        return [e[b, :n, :n] for b, n in enumerate(g["n"].tolist())]

    def forward(self, ingredients: torch.LongTensor, attn: torch.Tensor, attn_cls: torch.Tensor
                ) -> Dict[str, List[torch.Tensor]]:
        """Reference-compatible output (reference :377-399): three python lists of per-image
        tensors (views into the padded batch).  Costs one host sync for the sizes; use
        `instance_graph_padded` + `Matcher.forward_padded` to stay asynchronous."""
        return self._as_lists(self.instance_graph_padded(ingredients, attn, attn_cls))
