"""Shared 2-layer GCN that embeds instance graphs and class graphs
(reference schema_inference/graph/gnn.py).  Same module tree => same state-dict keys:
`embedding.weight`, `layers.{i}.g_conv.linear.{weight,bias}`, `layers.{i}.norm.{weight,bias}`,
`fc.{weight,bias}`.

Inference: every product runs on the matrix cores as split-fp16 MFMA GEMMs with the elementwise steps as
epilogues (csrc/sn_gcn.hip; `_forward_mfma`, `_forward_mfma_wide`), fp32-GEMM accuracy so the scores stay
within the 1e-5 budget; other configurations use the library GEMMs with HIP kernels in between.
Training (any input or parameter requires grad): the same maths as differentiable torch ops, with the
adjacency products adj @ X - forward and both gradients, the 27 GFLOP per step of the class graphs - on the
same MFMA GEMM through `ops.edges_adj_matmul` (an autograd.Function); layer 1's Linear folded into the embedding table as in
inference (`ops.gather_adj_matmul`), the other Linear layers and `fc` through `ops.linear_mfma`: no library GEMM in an iteration.
"""
import math
import os
from typing import Callable, Optional

import torch
import torch.nn as nn

from cpp_extension import ops

def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


_ACTIVATIONS = {
    "relu": nn.ReLU, "gelu": nn.GELU, "glu": nn.GLU, "swish": nn.SiLU, "sigmoid": nn.Sigmoid,
    "hard_sigmoid": nn.Hardsigmoid, "none": nn.Identity,
}


def get_activation_fn(name: str) -> Callable[[torch.Tensor], torch.Tensor]:
    """reference models/layers/__init__.py:16-26"""
    return _ACTIVATIONS[name]()


class _LinearPerGraphWeightGrad(torch.autograd.Function):
    """y = x W^T + b on x [G, n, in] with the weight gradient summed graph by graph.

    The library computes dW = dY^T X of the flattened [G n, out] x [G n, in] operands as ONE [out, in] product with an inner
    length of G n (103 k rows for the 101 class graphs of the Caltech configuration): 256 tiles of 16 x 16, each walking
    the whole inner length - 2.9 ms per layer on an MI355X, 22 % of a training iteration.  The same sum taken as G products
    of inner length n and one reduction over G is a batched GEMM with G x (out / tile) x (in / tile) workgroups: 0.1 ms."""

    @staticmethod
    @ops._amp_fwd
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @ops._amp_bwd
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy.matmul(weight)
        if ctx.needs_input_grad[1]:
            G, out_f, in_f = dy.shape[0], dy.shape[2], x.shape[2]
            # the [G, out, in] temporary of the per-graph products is 4 G out in bytes (config [3]: 1000 class graphs of width
            # 1024 = 4.2 GB): above _DW_TEMP_BYTES the graphs are taken in groups, each group's sum added into one [out, in] buffer
            group = max(1, min(G, _DW_TEMP_BYTES // max(1, 4 * out_f * in_f)))
            if group >= G:
                dw = torch.bmm(dy.transpose(1, 2), x).sum(dim=0)
            else:
                dw = torch.zeros(out_f, in_f, dtype=x.dtype, device=x.device)
                for g0 in range(0, G, group):
                    dw += torch.bmm(dy[g0:g0 + group].transpose(1, 2), x[g0:g0 + group]).sum(dim=0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 1))
        return dx, dw, db


_DW_TEMP_BYTES = 256 << 20     # largest [group, out, in] fp32 temporary of _LinearPerGraphWeightGrad.backward


def _linear(lin, x, extents=None):
    """`lin(x)`; for a batch of graphs on the GPU under autograd: on the matrix cores (ops.linear_mfma: forward and both gradients
    as split-fp16 MFMA GEMMs, no library GEMM in a training iteration), or - SN_LINEAR_MFMA=0 - the library's GEMMs with the
    per-graph weight gradient above."""
    if isinstance(lin, nn.Linear) and x.is_cuda and torch.is_grad_enabled() and lin.weight.requires_grad and x.dim() in (2, 3) and x.shape[0] > 0:
        if (x.dtype == torch.float32 and lin.in_features % 16 == 0 and lin.out_features % 16 == 0 and os.environ.get("SN_GCN_MFMA", "1") != "0"
                and os.environ.get("SN_LINEAR_MFMA", "1") != "0"):
            return ops.linear_mfma(x, lin.weight, lin.bias, extents if x.dim() == 3 else None)
        if x.dim() == 3 and x.shape[0] > 1 and os.environ.get("SN_LINEAR_PER_GRAPH_DW", "1") != "0":
            return _LinearPerGraphWeightGrad.apply(x, lin.weight, lin.bias)
    return lin(x)


class GraphConv(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, identity_proj: bool = False):
        super().__init__()
        assert not identity_proj or in_dim == out_dim
        self.linear = nn.Identity() if identity_proj else nn.Linear(in_dim, out_dim)
        if isinstance(self.linear, nn.Linear):
            nn.init.xavier_uniform_(self.linear.weight)
            nn.init.normal_(self.linear.bias)

    @staticmethod
    def adjacency(edges: torch.Tensor) -> torch.Tensor:
        """(E + E^T) / 2 + I   (reference gnn.py:27-30), differentiable form."""
        eye = torch.eye(edges.shape[-1], dtype=edges.dtype, device=edges.device)
        return (edges + edges.transpose(1, 2)) / 2 + eye

    def forward(self, edges: torch.Tensor, feat: torch.Tensor, adj: torch.Tensor = None, adj_planes=None, sum_edge_grads=False) -> torch.Tensor:
        if adj is None and adj_planes is None:
            adj = self.adjacency(edges)
        if adj_planes is not None:      # training on the matrix cores: adj @ feat and both of its gradients as split-fp16 MFMA GEMMs
            if adj is None:             # ... with the adjacency built straight from the edges as fp16 planes (never dense)
                # (compacted class graphs: the operand holds the kept vertices only, every product takes their counts as extents)
                ext = adj_planes.compact[1] if getattr(adj_planes, "compact", None) is not None else None
                return _linear(self.linear, ops.edges_adj_matmul(edges, feat, adj_planes, sum_edge_grads), ext)
            return _linear(self.linear, ops.sym_adj_matmul(adj, feat, adj_planes))
        return _linear(self.linear, torch.bmm(adj, feat))


class Layer(nn.Module):
    def __init__(self, emb_dim: int, activation: str, identity_proj: bool = False):
        super().__init__()
        self.g_conv = GraphConv(emb_dim, emb_dim, identity_proj)
        self.norm = nn.LayerNorm(emb_dim)
        self.activation = get_activation_fn(activation)
        self._is_relu = activation == "relu"
        self._is_none = activation == "none"

    def forward(self, edges: torch.Tensor, feat: torch.Tensor, feat_mask: torch.BoolTensor = None,
                adj: torch.Tensor = None, n_valid: torch.Tensor = None, fused: bool = False, adj_planes=None, sum_edge_grads=False):
        return self.post(self.g_conv(edges, feat, adj, adj_planes, sum_edge_grads), feat_mask, n_valid, fused)

    def post(self, feat: torch.Tensor, feat_mask: torch.BoolTensor = None, n_valid: torch.Tensor = None, fused: bool = False):
        """what follows the graph convolution: pad rows -> 0, LayerNorm, activation (reference gnn.py:43-46)"""
        if fused and (self._is_relu or self._is_none) and feat.is_contiguous():
            return ops.mask_layernorm_act_(feat, self.norm.weight, self.norm.bias, self.norm.eps,
                                           n_valid=n_valid, relu=self._is_relu)
        if (feat.is_cuda and feat.dim() == 3 and feat.dtype == torch.float32 and torch.is_grad_enabled() and (self._is_relu or self._is_none)
                and feat.shape[-1] <= 1024 and (feat_mask is None or n_valid is not None) and self.norm.elementwise_affine
                and self.norm.bias is not None and os.environ.get("SN_LN_ACT_FUSED", "1") != "0"):
            # training on the GPU: mask, LayerNorm and activation as one differentiable op (one pass forward, one back)
            return ops.mask_layernorm_act(feat, self.norm.weight, self.norm.bias, self.norm.eps,
                                          n_valid=n_valid if feat_mask is not None else None, relu=self._is_relu)
        if feat_mask is not None:
            feat = feat.masked_fill(feat_mask[..., None], 0)
        return self.activation(self.norm(feat))


class GNN(nn.Module):
    def __init__(self, num_codes: int, embed_dim: int, num_layers: int, identity_proj: bool = False,
                 activation: str = "relu"):
        super().__init__()
        self.num_codes = num_codes
        self.embed_dim = embed_dim
        self.num_layers = num_layers
        self.embedding = nn.Embedding(num_codes + 1, embed_dim, padding_idx=num_codes)
        self.layers = nn.ModuleList([Layer(embed_dim, activation, identity_proj) for _ in range(num_layers)])
        self._prepared = None
        self.fc = nn.Linear(embed_dim, embed_dim)
        nn.init.normal_(self.fc.weight)
        nn.init.zeros_(self.fc.bias)
        with torch.no_grad():
            nn.init.trunc_normal_(self.embedding.weight[:num_codes])

    def _cached_sort(self, ids: torch.Tensor, rows: int):
        """(order, seg) of an index tensor that is a module's buffer / Parameter (the class graphs' words,
        `SchemaNet.class_ingredients`: the same in every iteration): its sort is taken once per version of the tensor and kept ON
        it; None for any other index tensor, or when a sort would be needed while a stream capture is running."""
        if not (isinstance(ids, nn.Parameter) and not ids.requires_grad and ids.is_cuda and os.environ.get("SN_EMBED_SORTED", "1") != "0"):
            return None
        srt = getattr(ids, "_sn_sorted", None)
        if srt is None or srt[0] != (ids._version, rows):
            if torch.cuda.is_current_stream_capturing():
                return None                                  # (a sort cannot be captured: whoever captures warms up first)
            with torch.no_grad():
                srt = ((ids._version, rows),) + ops.sorted_ids_of(ids.detach().clamp(0, rows - 1), rows)
            ids._sn_sorted = srt
        return srt[1], srt[2]

    def _embed(self, ids: torch.Tensor) -> torch.Tensor:
        """`self.embedding(ids)`.  Training on the GPU with an index tensor whose sort is cached: the backward pass is one
        gather-sum (ops.embedding_sorted) instead of the library's sort of the 103 k ids in every iteration."""
        w = self.embedding.weight
        if w.is_cuda and w.dtype == torch.float32 and w.requires_grad and torch.is_grad_enabled() and w.shape[1] % 4 == 0:
            srt = self._cached_sort(ids, w.shape[0])
            if srt is not None:
                return ops.embedding_sorted(w, ids.detach(), srt[0], srt[1], self.embedding.padding_idx)
        return self.embedding(ids)

    def _differentiable(self, *tensors) -> bool:
        if not torch.is_grad_enabled():
            return False
        return any(t.requires_grad for t in tensors if t is not None) or any(p.requires_grad for p in self.parameters())

    def _mfma_ok(self) -> bool:
        """The split-fp16 MFMA path (csrc/sn_gcn.hip) covers two layers with a Linear projection and ReLU / no
        activation: embed_dim 256 (the shipped CIFAR / Caltech configurations) with every elementwise step fused into
        the GEMM epilogues, other widths that are a multiple of 16 (1024 in the ImageNet yaml) with the same GEMMs and
        the LayerNorm / pooling kernels of the library route in between."""
        if os.environ.get("SN_GCN_MFMA", "1") == "0" or self.embed_dim % 16 != 0 or len(self.layers) != 2:
            return False
        return all(isinstance(l.g_conv.linear, nn.Linear) and (l._is_relu or l._is_none) for l in self.layers)

    def masks_adjacency(self, edges: torch.Tensor) -> bool:
        """True when `forward(..., n_valid=...)` never reads edges outside a graph's own corner (the MFMA routes build
        the adjacency operand with sn_gcn_adjacency_planes_masked)."""
        return bool(edges.is_cuda and self._mfma_ok() and not self._differentiable(edges))

    def prepare(self):
        """The parameter-only operands of the MFMA path: layer 1's Linear folded into the embedding
        table ((M+1) x E GEMM) and W2 as fp16 hi/lo planes.  A forward pass embeds the instance graphs
        AND the class graphs with the same weights (Matcher), so it computes them once and hands them
        to both `forward(..., prepared=...)` calls; None when the MFMA path does not apply."""
        if not self._mfma_ok() or not self.embedding.weight.is_cuda or self._differentiable():
            return None
        l1, l2 = self.layers
        # weight-only operands: kept until one of the four weights changes (data_ptr / _version, like the packed codebook
        # of S1; `p.data.copy_` writes need `invalidate_prepared()`), so a forward pass launches no library GEMM
        srcs = (self.embedding.weight, l1.g_conv.linear.weight, l2.g_conv.linear.weight, self.fc.weight,
                l1.norm.weight, l1.norm.bias, l2.norm.weight, l2.norm.bias,       # (the LayerNorm parameters: the operand scales)
                l1.g_conv.linear.bias, l2.g_conv.linear.bias)                     # (... and the biases: the isolated-vertex table)
        fused_gather = self.embed_dim % 256 == 0 and os.environ.get("SN_GCN_GATHER_FUSED", "1") == "1"
        fused_linear = self.embed_dim == 256 and os.environ.get("SN_GCN_FUSE_LINEAR", "1") == "1"
        key = tuple((t.data_ptr(), t._version, t.device) for t in srcs) + (fused_gather, fused_linear)
        if getattr(self, "_prepared", None) is not None and self._prepared[0] == key:
            return self._prepared[1]
        table = ops.gcn_gemm(ops.split_planes(self.embedding.weight), ops.split_planes(l1.g_conv.linear.weight), 1, want_c=True)["c"][0]
        # Power-of-two scales of the split-fp16 operands (csrc/sn_gcn.hip, "What hi + lo holds"): device scalars, computed
        # here once per weight version without a host synchronisation.  Weight-only operands: from their largest
        # magnitude.  Operands written by a GEMM epilogue: from a BOUND on what the epilogue can produce - a LayerNorm
        # output of width E is at most sqrt(E - 1) standard deviations from its mean (16 at E = 256, 32 at E = 1024: the wide
        # route uses the same scales), so |H| <= ceil(sqrt(E - 1)) max|gamma| + max|beta|, and |W2 . H^T| <= max_o |W2[o, :]|_1 x that.
        ln_sigmas = float(math.ceil(math.sqrt(max(1, self.embed_dim - 1))))

        def ln_bound(norm):
            return ln_sigmas * norm.weight.detach().abs().amax() + norm.bias.detach().abs().amax()
        w2 = l2.g_conv.linear.weight.detach()
        h1_bound = ln_bound(l1.norm)
        out = {"table": table, "table_scale": ops.pow2_scale(table), "w2": ops.split_planes(w2),
               "h1_scale": ops.pow2_scale(h1_bound), "h2_scale": ops.pow2_scale(ln_bound(l2.norm)),
               "zt2_scale": ops.pow2_scale(h1_bound * w2.abs().sum(dim=1).amax()),
               "fc_t": self.fc.weight.detach().t().contiguous()}                      # [E, E_out]: ops.pool_fc reads whole lines of it
        # What an ISOLATED vertex of word w contributes to its graph's pooled feature, per unit of node weight: with no edge
        # its adjacency row is the identity, so both products pass its own row through - H2 = act(LN2(W2 act(LN1(table[w] +
        # b1)) + b2)), a function of the word only.  The pruned vertices of a trained IR-Atlas are such nodes
        # (schema_net.py:152-166 zeroes their rows and columns): the compacted class branch (forward(..., compact=)) leaves
        # them out of its products and adds sum_i w_i iso[word_i] instead.  [M + 1, E], once per weight version.
        with torch.no_grad():
            h = torch.nn.functional.layer_norm(table + l1.g_conv.linear.bias, (self.embed_dim,), l1.norm.weight, l1.norm.bias, l1.norm.eps)
            h = torch.relu(h) if l1._is_relu else h
            h = torch.nn.functional.linear(h, l2.g_conv.linear.weight, l2.g_conv.linear.bias)
            h = torch.nn.functional.layer_norm(h, (self.embed_dim,), l2.norm.weight, l2.norm.bias, l2.norm.eps)
            out["iso"] = (torch.relu(h) if l2._is_relu else h).contiguous()
        if fused_gather:
            # layer 1 gathers its B operand inside the GEMM: 170 MB less HBM traffic and 20 us less kernel time per step
            # (+4 % with four in-line steps in flight, +3 % one step at a time; SN_GCN_GATHER_FUSED=0 = the
            # separate gather kernel)
            out["table_planes"] = ops.table_planes(table, out["table_scale"])
        if fused_linear:
            # layer 2's Linear runs in the epilogue of layer 1's product (H1 never leaves the workgroup): its weight with
            # the columns in the epilogue's feature order
            out["w2_next"] = ops.next_layer_weight_planes(l2.g_conv.linear.weight)
        self._prepared = (key, out)
        return out

    def train_folds(self, device_is_cuda: bool = True) -> bool:
        """True when `forward` under autograd takes the route with layer 1's Linear folded into the embedding table"""
        first = self.layers[0] if len(self.layers) else None
        return bool(device_is_cuda and first is not None and isinstance(first.g_conv.linear, nn.Linear) and self.embed_dim % 16 == 0
                    and self.embedding.weight.is_cuda and self.embedding.weight.dtype == torch.float32
                    and os.environ.get("SN_GCN_MFMA", "1") != "0" and os.environ.get("SN_TRAIN_FOLD", "1") != "0")

    def train_table(self):
        """embedding.weight @ W1^T with autograd (ops.linear_mfma): the folded table of the training route, for a caller that runs
        several passes in one iteration (`forward(..., prepared={"train_table": t})`: Matcher embeds the instance graphs and the class
        graphs with the same weights)"""
        return ops.linear_mfma(self.embedding.weight, self.layers[0].g_conv.linear.weight)

    def invalidate_prepared(self):
        self._prepared = None

    @staticmethod
    def _compacted(nodes, ingredients, compact, iso):
        """class graphs whose operand holds their kept vertices only (SchemaNet.get_atlas(fused_adjacency="compact")):
        -> (ids and node weights in the operand's vertex order - weights zero beyond a graph's extent -, the per-graph extents,
        and the buffer of the pooled partial sums [G, row tiles + 1, E] whose LAST slot already holds the isolated vertices' share)"""
        perm, n_kept = compact
        G, n = ingredients.shape
        pooled = torch.empty((G, (n + 127) // 128 + 1, iso.shape[1]), dtype=torch.float32, device=iso.device)
        ids_c, w_c, _ = ops.class_compact(perm, n_kept, nodes, ingredients, iso, pooled_slot=pooled[:, -1, :])
        return ids_c, w_c, n_kept, pooled

    def _forward_mfma(self, nodes, edges, ingredients, n_valid, divisor, adj=None, prepared=None, compact=None):
        """Inference on the matrix cores: three GEMM launches per call, every elementwise step an
        epilogue (bias, pad-row mask, LayerNorm, ReLU, hi/lo split, node-weighted pooling).
            H1     = act(LN1(adj @ (Emb @ W1^T)[ids] + b1))
            Zt2    = W2 @ H1^T                                   (so that adj @ Zt2^T = adj @ H1 @ W2^T)
            pooled = sum_i nodes_i * act(LN2(adj @ Zt2^T + b2))_i
        """
        G, n = ingredients.shape
        E = self.embed_dim
        l1, l2 = self.layers
        # `divisor` (when given) is the largest vertex count of the batch, on the device: nothing beyond it
        # (rounded up to the block sizes) is produced or multiplied - no host synchronisation needed
        ext = divisor if (divisor is not None and torch.is_tensor(divisor)) else None
        if prepared is None:
            prepared = self.prepare()
        # per-graph extents (see below): decided here because the adjacency operand is then produced per graph as well
        graph_ext = bool(adj is None and compact is None and ext is not None and n_valid is not None and n_valid.numel() == G and G > 1
                         and n_valid.dtype == torch.int32 and "w2_next" in prepared and os.environ.get("SN_GCN_GRAPH_EXTENTS", "1") != "0")
        if adj is None:
            adj = ops.gcn_adjacency_planes(edges, extent=ext, n_valid=n_valid.contiguous() if graph_ext else n_valid, per_graph=graph_ext)     # A  [G, n, n]
        pooled_buf = None
        if compact is not None:                          # (class graphs of a pruned atlas: per-graph extents)
            ingredients, nodes, ext, pooled_buf = self._compacted(nodes, ingredients, compact, prepared["iso"])
            n_valid = ext
        if "table_planes" in prepared and adj.kpad <= 1024:
            # Bt[g, f, j] = table[ids[g, j], f] gathered inside the kernel (no [G, E, n] copy through HBM)
            t_hi, t_lo = prepared["table_planes"]
            zt1, b_table = None, (t_hi, t_lo, ingredients.contiguous(), prepared["table_scale"])
        else:
            zt1, b_table = ops.gcn_gather_planes(prepared["table"], ingredients, extent=ext, scale=prepared["table_scale"]), None     # Bt [G, E, n]
        # Extents of the products.  A batch of graphs with their own vertex counts (instance graphs: ~113 vertices each, the
        # largest of a batch ~160) takes them PER GRAPH (round 5; the compacted class graphs always did): a graph of at most 128
        # vertices costs one row tile and its own k-stages where the batch maximum made it two tiles - the rows and stages beyond a
        # graph's count are pad rows (zeroed, gnn.py:43-45) times zero adjacency columns, and nothing downstream reads them.
        gext = n_valid.contiguous() if graph_ext else ext
        if "w2_next" in prepared:
            zt2 = ops.gcn_gemm(adj, zt1, G, bias=l1.g_conv.linear.bias,
                               layernorm=(l1.norm.weight, l1.norm.bias, l1.norm.eps), relu=l1._is_relu,
                               rows_valid=n_valid, want_planes=n, m_extent=gext, k_extent=gext, b_table=b_table,
                               next_w=prepared["w2_next"], h_scale=prepared["h1_scale"],
                               out_scale=prepared["zt2_scale"])["planes"]            # W2 @ H1^T [G, E, n], H1 never stored
        else:
            h1 = ops.gcn_gemm(adj, zt1, G, bias=l1.g_conv.linear.bias,
                              layernorm=(l1.norm.weight, l1.norm.bias, l1.norm.eps), relu=l1._is_relu,
                              rows_valid=n_valid, want_planes=E, m_extent=ext, k_extent=ext, b_table=b_table,
                              out_scale=prepared["h1_scale"])["planes"]             # [G, n, E]
            zt2 = ops.gcn_gemm(prepared["w2"], h1, G, want_planes=n, out_scale=prepared["zt2_scale"])["planes"]      # A = W2 planes [1, E, E] -> [G, E, n]
        pooled = ops.gcn_gemm(adj, zt2, G, bias=l2.g_conv.linear.bias,
                              layernorm=(l2.norm.weight, l2.norm.bias, l2.norm.eps), relu=l2._is_relu,
                              rows_valid=n_valid, pool_w=nodes, m_extent=gext, k_extent=gext,
                              pooled_out=pooled_buf)["pooled"]   # [G, row tiles (+ 1: the isolated vertices' share), E]
        return ops.pool_fc(pooled, divisor if divisor is not None else n, self.fc.weight, self.fc.bias, weight_t=prepared.get("fc_t"))

    def _forward_mfma_wide(self, nodes, edges, ingredients, n_valid, divisor, adj=None, prepared=None, compact=None):
        """embed_dim != 256: a row of the result spans several 256-column tiles, so LayerNorm cannot be an epilogue of
        the GEMM.  Same three products on the matrix cores (split fp16, fp32-GEMM accuracy), fp32 results; mask + LayerNorm +
        activation run inside the pass that splits H1 for the next product, and inside the pooling pass for H2."""
        G, n = ingredients.shape
        l1, l2 = self.layers
        ext = divisor if (divisor is not None and torch.is_tensor(divisor)) else None
        if adj is None:
            adj = ops.gcn_adjacency_planes(edges, extent=ext, n_valid=n_valid)
        if prepared is None:
            prepared = self.prepare()
        if compact is not None:
            raise RuntimeError("compacted class graphs need the fully fused route (embed_dim 256)")
        if "table_planes" in prepared and adj.kpad <= 1024:
            # Bt[g, f, j] = table[ids[g, j], f] gathered inside the product, each 256-column tile its own slice of the table rows
            # (round 4: the [G, E, n] planes - 2.1 GB at config [3] - are neither written nor read back)
            t_hi, t_lo = prepared["table_planes"]
            zt1, b_table = None, (t_hi, t_lo, ingredients.contiguous(), prepared["table_scale"])
        else:
            zt1, b_table = ops.gcn_gather_planes(prepared["table"], ingredients, extent=ext, scale=prepared["table_scale"]), None   # Bt [G, E, n]
        c1 = ops.gcn_gemm(adj, zt1, G, bias=l1.g_conv.linear.bias, rows_valid=n_valid, want_c=True, zero_c=ext is not None,
                          m_extent=ext, k_extent=ext, b_table=b_table)["c"]                          # [G, n, E]
        # mask + LayerNorm + activation and the hi / lo split of H1 in one pass (H1 itself is never stored)
        h1 = ops.layernorm_split_planes(c1, l1.norm.weight, l1.norm.bias, l1.norm.eps, n_valid=n_valid, relu=l1._is_relu,
                                        scale=prepared["h1_scale"])
        zt2 = ops.gcn_gemm(prepared["w2"], h1, G, want_planes=n, out_scale=prepared["zt2_scale"])["planes"]   # [G, E, n]
        c2 = ops.gcn_gemm(adj, zt2, G, bias=l2.g_conv.linear.bias, rows_valid=n_valid, want_c=True, zero_c=ext is not None,
                          m_extent=ext, k_extent=ext)["c"]
        pooled = ops.layernorm_weighted_pool(c2, l2.norm.weight, l2.norm.bias, l2.norm.eps, nodes, n_valid=n_valid, relu=l2._is_relu,
                                             divisor_dev=divisor)                                    # (nor H2)
        return ops.pool_fc(pooled, 1.0, self.fc.weight, self.fc.bias, weight_t=prepared.get("fc_t"))    # (the final Linear without a library GEMM)

    def forward(self, nodes: torch.Tensor, edges: torch.Tensor, ingredients: torch.LongTensor,
                feat_mask: torch.BoolTensor = None, n_valid: torch.Tensor = None,
                divisor: Optional[torch.Tensor] = None, adjacency=None, prepared=None, compact=None) -> torch.Tensor:
        """nodes [G, n], edges [G, n, n], ingredients [G, n] -> graph feature [G, embed_dim].

        feat_mask (bool [G, n], True = padding) is the reference argument (gnn.py:78-98).
        n_valid (int32 [G]) is its compact device form; `divisor` (int32 [1] device tensor)
        replaces the padded length in the mean pooling (gnn.py:96) so that graphs padded to a
        fixed n_pad give the same result as graphs padded to the batch maximum.
        """
        if adjacency is not None:        # prebuilt (E + E^T)/2 + I planes (SchemaNet.get_atlas(fused_adjacency=True))
            if not (nodes.is_cuda and self._mfma_ok()) or self._differentiable(nodes):
                raise RuntimeError("adjacency planes need the inference MFMA path (2 Linear layers, embed_dim a multiple of 16, no autograd)")
            run = self._forward_mfma if self.embed_dim == 256 else self._forward_mfma_wide
            return run(nodes, None, ingredients, n_valid, divisor, adj=adjacency, prepared=prepared, compact=compact)
        fused = nodes.is_cuda and not self._differentiable(nodes, edges)
        if n_valid is None and feat_mask is not None:
            n_valid = (~feat_mask).sum(dim=1).to(torch.int32)   # masks are suffix masks (match.py:48-51)
        if fused and self._mfma_ok():
            run = self._forward_mfma if self.embed_dim == 256 else self._forward_mfma_wide
            return run(nodes, edges, ingredients, n_valid, divisor, prepared=prepared)
        if feat_mask is None and n_valid is not None and not fused:
            feat_mask = torch.arange(nodes.shape[1], device=nodes.device)[None, :] >= n_valid[:, None]
        # training on the GPU: the adjacency is only ever built as fp16 planes, one pass over the edges (ops.edges_adj_matmul)
        train_mfma = not fused and edges.is_cuda and self.embed_dim % 16 == 0 and os.environ.get("SN_GCN_MFMA", "1") != "0"
        adj = None if train_mfma else (ops.gcn_adjacency(edges) if fused else GraphConv.adjacency(edges))
        layers = list(self.layers)
        first = layers[0] if layers else None
        feat = None
        if fused and first is not None and isinstance(first.g_conv.linear, nn.Linear) and (first._is_relu or first._is_none):
            # layer 1 re-associated: (adj @ Emb[ids]) @ W^T + b == adj @ (Emb @ W^T)[ids] + b, so the
            # [G*n, E] x [E, E] Linear becomes one [(M+1), E] x [E, E] GEMM shared by every graph
            lin = first.g_conv.linear
            table = torch.nn.functional.linear(self.embedding.weight, lin.weight)
            feat = torch.baddbmm(lin.bias, adj, torch.nn.functional.embedding(ingredients, table))
            feat = ops.mask_layernorm_act_(feat, first.norm.weight, first.norm.bias, first.norm.eps,
                                           n_valid=n_valid, relu=first._is_relu)
            layers = layers[1:]
        fold = bool(feat is None and train_mfma and first is not None and isinstance(first.g_conv.linear, nn.Linear) and ingredients.dtype == torch.int64
                    and self.embedding.weight.dtype == torch.float32 and edges.dtype == torch.float32 and os.environ.get("SN_TRAIN_FOLD", "1") != "0")
        pooled_iso, compact_sort = None, None
        if fold and compact is not None and all(isinstance(l.g_conv.linear, nn.Linear) for l in layers):
            # Training with COMPACTED class graphs (round 5): a vertex under the prune threshold has no edge (schema_net.py:152-166 zeroes its
            # row and column), so its adjacency row is the identity and its feature a function of its word only.  The products run on the
            # kept vertices of every class (perm: kept first; their counts are the extents of every product, pad rows masked as in the
            # instance graphs), the pruned ones enter the pooled class feature through the per-word table `iso` - with autograd, so
            # their share of the gradients of the GNN's weights and of the vertex weights is kept.
            perm, n_kept = compact
            perm_l = perm.long()
            kept = torch.arange(nodes.shape[1], device=nodes.device)[None, :] < n_kept[:, None]
            # (the cached sort of the stored ids carried over to the permuted order: position (g, i) of the stored order is row
            # (g, inv[g, i]) of the compacted one - the table's gradient stays one gather-sum, no sort per iteration)
            compact_sort = None
            srt0 = self._cached_sort(ingredients, self.embedding.weight.shape[0])
            if srt0 is not None:
                n_ = ingredients.shape[1]
                inv = torch.empty_like(perm_l).scatter_(1, perm_l, torch.arange(n_, device=perm.device).expand_as(perm_l))
                order = srt0[0]
                compact_sort = ((order // n_) * n_ + inv.reshape(-1)[order], srt0[1])
            ingredients = ingredients.gather(1, perm_l)
            w_all = nodes.gather(1, perm_l)
            nodes = w_all * kept.to(w_all.dtype)
            feat_mask, n_valid = ~kept, n_kept
            adj_planes = ops.gcn_adjacency_planes_compact(_contig(edges.detach()), perm, n_kept)
            table = prepared.get("train_table") if isinstance(prepared, dict) else None
            if table is None:
                table = ops.linear_mfma(self.embedding.weight, first.g_conv.linear.weight)
            prepared = {"train_table": table}
            # (the padding word's row of the table takes no gradient on this path either: nn.Embedding keeps embedding.weight[padding_idx].grad
            # at 0 - reference gnn.py:63-67 - and the uncompacted route does so through gather_adj_matmul's `pad`; a class vertex whose
            # word IS the padding index reaches this table only here: ADVICE r05)
            pad = self.embedding.padding_idx
            table_iso = table
            if pad is not None and 0 <= pad < table.shape[0]:
                table_iso = table.clone()
                table_iso[pad] = table[pad].detach()
            h = first.post((table_iso + first.g_conv.linear.bias)[None], None, None, False)[0]               # an isolated vertex of word w, layer by layer
            for l in layers[1:]:
                h = l.post(_linear(l.g_conv.linear, h)[None], None, None, False)[0]
            w_sum = torch.zeros((nodes.shape[0], table.shape[0]), dtype=w_all.dtype, device=nodes.device)
            w_sum = w_sum.scatter_add(1, ingredients.clamp(0, table.shape[0] - 1), w_all * (~kept).to(w_all.dtype))   # [G, words]: pruned weight per word
            pad = (-table.shape[0]) % 16
            pooled_iso = ops.linear_mfma(torch.nn.functional.pad(w_sum, (0, pad)), torch.nn.functional.pad(h.t(), (0, pad)).contiguous())
        else:
            adj_planes = ops.gcn_adjacency_planes(_contig(edges.detach())) if train_mfma else None      # shared by the layers (and by their backward passes)
        if fold:
            # training, layer 1 re-associated the same way: the Linear of the first convolution acts on the embedding TABLE (a [M + 1, E] x
            # [E, E] product with autograd) and the graph product gathers its operand from that table as fp16 planes - no [G n, E] x [E, E]
            # GEMM in the forward or the backward pass, no fp32 embedding of the graphs
            lin = first.g_conv.linear
            table = prepared.get("train_table") if isinstance(prepared, dict) else None     # (Matcher: one table for both passes of an iteration)
            if table is None:
                table = ops.linear_mfma(self.embedding.weight, lin.weight)
            srt = self._cached_sort(ingredients, table.shape[0]) if pooled_iso is None else compact_sort
            feat = ops.gather_adj_matmul(edges, table, ingredients.detach(), lin.bias, adj_planes, srt, self.embedding.padding_idx, sum_edge_grads=True)
            feat = first.post(feat, feat_mask, n_valid, fused)
            layers = layers[1:]
        if feat is None:
            feat = self._embed(ingredients)
        for layer in layers:
            # (the layers run in sequence on one `edges` / `adj_planes`: their edge gradients are summed before autograd sees them)
            feat = layer(edges, feat, feat_mask, adj=adj, n_valid=n_valid, fused=fused, adj_planes=adj_planes, sum_edge_grads=train_mfma)
        if fused:
            pooled = ops.weighted_pool(feat, nodes, divisor)
        elif (feat.is_cuda and feat.dtype == torch.float32 and feat.shape[-1] % 4 == 0 and (divisor is None or (torch.is_tensor(divisor) and divisor.dtype == torch.int32))
              and os.environ.get("SN_POOL_FUSED", "1") != "0"):
            pooled = ops.weighted_pool_autograd(feat, nodes, divisor)         # training: one pass forward, one back
        else:
            pooled = (feat * nodes[..., None]).sum(dim=1)
            pooled = pooled / (divisor.to(pooled.dtype) if divisor is not None else feat.shape[1])
        if pooled_iso is not None:                                           # (compacted class graphs: the pruned vertices' share)
            pooled = pooled + pooled_iso / (divisor.to(pooled.dtype) if divisor is not None else float(feat.shape[1]))
        return _linear(self.fc, pooled)
