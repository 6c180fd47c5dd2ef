"""Tensor-level wrappers of the C ABI: allocate outputs, pass data_ptr()s + the current HIP
stream, raise RuntimeError on a non-zero status.  All tensors are CUDA tensors here; the
reference-compatible shims (CPU tensors in, lists out) live in cpp_extension/__init__.py.
"""
import os
from ctypes import byref, c_void_p, pointer

import torch
from torch.amp import custom_bwd, custom_fwd

# Autocast policy of the autograd.Functions below (reference trainer: worker_schema_net.py:128-143, `use_amp`): the HIP
# kernels compute in fp32 and take raw pointers, so under `torch.autocast` their floating-point inputs are cast to fp32 and
# the op itself runs with autocast off (forward and backward) - the policy torch applies to its own fp32-only ops.
_amp_fwd = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = custom_bwd(device_type="cuda")

from . import _native as N


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


def _check_dev(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("expected CUDA tensors (the HIP path has no CPU fallback)")
        dev = dev or t.device
        if t.device != dev:
            raise RuntimeError(f"tensors on different devices: {t.device} vs {dev}")
    return dev


# ------------------------------------------------------------------------------- S1
class PackedCodebook:
    """fp16 MFMA-fragment image + fp64 norms of a codebook (sn_codebook_prepare).  Rebuilt when
    the codebook tensor changes (data_ptr / _version / device / shape).  Writes that bypass the version
    counter (`weight.data.copy_(...)`, raw pointer writes) are not seen: call `invalidate()` after them
    (`Discretization.load_state_dict` / `initial_vocabulary` do)."""

    def __init__(self):
        self.key = None
        self.buf = None
        self.codebook = None

    def invalidate(self):
        self.key = None

    def get(self, codebook):
        cb = _f32c(codebook.detach())
        key = (cb.data_ptr(), codebook._version, cb.device, tuple(cb.shape))
        if key != self.key:
            lib = N.require_gpu()
            M, D = cb.shape
            nbytes = lib.sn_codebook_pack_bytes(M, D)
            if nbytes == 0:
                raise RuntimeError(f"unsupported codebook shape M={M}, D={D} (need D % 32 == 0, D <= 1024, M <= 65536)")
            buf = torch.empty(nbytes, dtype=torch.uint8, device=cb.device)
            with torch.cuda.device(cb.device):
                N.check(lib.sn_codebook_prepare(N.ptr(cb), M, D, N.ptr(buf), N.stream_ptr(cb.device)), "sn_codebook_prepare")
            self.key, self.buf, self.codebook = key, buf, cb
        return self.codebook, self.buf


class DeferredRerank:
    """What `assign_words(..., defer=True)` hands to the consumer of the word ids: the screen has run, `out` holds its
    words (final wherever it could prove them), the flag words and candidate records of the undecided tokens are in
    `ws`.  `instance_graph(..., rerank=handle)` finishes them inside its row phase (no re-rank launch between S1 and
    the graph); `finish()` runs the stand-alone re-rank instead (a consumer that cannot fuse it).  Either way the ids are
    those of mode 0, bit for bit."""

    def __init__(self, tokens, bf16, codebook, packed, out, ws, ws_bytes):
        self.tokens, self.bf16, self.codebook, self.packed, self.out, self.ws, self.ws_bytes = tokens, bf16, codebook, packed, out, ws, ws_bytes
        self.done = False

    def args_for(self, ingredients):
        """sn_rerank_args for a [B, L] view of `out` (the ids as the graph kernel sees them), or None when `ingredients`
        is not such a view"""
        if self.done or ingredients.data_ptr() != self.out.data_ptr():
            return None
        n_outer, n_inner, D = self.tokens.shape
        if tuple(ingredients.shape) == (n_outer, n_inner) and ingredients.stride() == self.out.stride():
            axis = 0                                              # batch-first tokens
        elif tuple(ingredients.shape) == (n_inner, n_outer) and ingredients.stride() == (self.out.stride(1), self.out.stride(0)):
            axis = 1                                              # sequence-first tokens: the batch is the inner axis
        else:
            return None
        r = N.RerankArgs()
        r.x = self.tokens.data_ptr()
        r.x_stride_b, r.x_stride_l = self.tokens.stride(axis), self.tokens.stride(1 - axis)
        r.x_bf16 = int(self.bf16)
        r.tok_stride_b, r.tok_stride_l = (n_inner, 1) if axis == 0 else (1, n_inner)
        r.n_tokens = n_outer * n_inner
        r.codebook, r.packed = self.codebook.data_ptr(), self.packed.data_ptr()
        r.M, r.D = self.codebook.shape[0], D
        r.workspace = self.ws.data_ptr()
        r.ids = self.out.data_ptr()
        r.ids_stride_b, r.ids_stride_l = self.out.stride(axis), self.out.stride(1 - axis)
        return r

    def finish(self):
        if self.done:
            return
        lib = N.require_gpu()
        t, o, dev = self.tokens, self.out, self.out.device
        with torch.cuda.device(dev):
            fn = lib.sn_assign_words_bf16 if self.bf16 else lib.sn_assign_words
            N.check(fn(N.ptr(t), t.shape[0], t.shape[1], t.stride(0), t.stride(1), N.ptr(self.codebook), N.ptr(self.packed),
                       self.codebook.shape[0], t.shape[2], N.ptr(o), o.stride(0), o.stride(1), N.ptr(self.ws), self.ws_bytes, 3,
                       N.stream_ptr(dev)), "sn_assign_words (mode 3)")
        self.done = True


def assign_words(tokens, codebook, packed, out=None, mode=0, defer=False):
    """tokens: [n_outer, n_inner, D] view (last dim contiguous, any outer strides);
    returns int64 [n_outer, n_inner] word ids (or fills `out`, any strides).
    defer=True (mode 0 only): returns (ids, handle) - handle a `DeferredRerank` whose consumer finishes the tokens the
    fp16 screen could not decide, or None where the shape has no deferred form (the ids are final then)."""
    lib = N.require_gpu()
    dev = _check_dev(tokens, codebook, packed, out)
    bf16 = tokens.dtype == torch.bfloat16 and tokens.stride(-1) == 1      # consumed in place (half the token bytes)
    if not bf16 and (tokens.dtype != torch.float32 or tokens.stride(-1) != 1):
        tokens = tokens.to(torch.float32).contiguous()
    assert tokens.dim() == 3
    n_outer, n_inner, D = tokens.shape
    M = codebook.shape[0]
    assert codebook.shape[1] == D and codebook.is_contiguous() and codebook.dtype == torch.float32
    if out is None:
        out = torch.empty((n_outer, n_inner), dtype=torch.int64, device=dev)
    assert out.dtype == torch.int64 and tuple(out.shape) == (n_outer, n_inner)
    n_tok = n_outer * n_inner
    ws_bytes = lib.sn_assign_workspace_bytes(n_tok)
    ws = torch.empty(max(ws_bytes, 32), dtype=torch.uint8, device=dev)
    deferred = bool(defer) and int(mode) == 0 and n_tok > 0 and lib.sn_assign_defers(M, D) == 1 and os.environ.get("SN_S1_DEFER", "1") != "0"
    with torch.cuda.device(dev):
        fn = lib.sn_assign_words_bf16 if bf16 else lib.sn_assign_words
        N.check(fn(
            N.ptr(tokens), n_outer, n_inner, tokens.stride(0), tokens.stride(1), N.ptr(codebook), N.ptr(packed), M, D,
            N.ptr(out), out.stride(0), out.stride(1), N.ptr(ws), ws_bytes, 2 if deferred else int(mode), N.stream_ptr(dev)), "sn_assign_words")
    if defer:
        return out, (DeferredRerank(tokens, bf16, codebook, packed, out, ws, ws_bytes) if deferred else None)
    return out


def kmeans_update(tokens, ids, K, sorted_route=True):
    """M-step of Lloyd's k-means.  tokens [n_outer, n_inner, D] f32 view, ids int64 [n_outer, n_inner]
    (the output of assign_words).  -> (sums f32 [K, D], counts int64 [K]): per-centre member sums added in
    token order (bit-identical to SciPy's float32 update) and member counts.  sorted_route: a stable sort of
    the ids groups the tokens by centre first (sn_kmeans_update_sorted), otherwise every workgroup walks the id
    stream (sn_kmeans_update); same result."""
    lib = N.require_gpu()
    dev = _check_dev(tokens, ids)
    if tokens.dtype != torch.float32 or tokens.stride(-1) != 1:
        tokens = tokens.to(torch.float32).contiguous()
    assert tokens.dim() == 3 and ids.dtype == torch.int64 and tuple(ids.shape) == tuple(tokens.shape[:2])
    n_outer, n_inner, D = tokens.shape
    sums = torch.empty((K, D), dtype=torch.float32, device=dev)
    counts = torch.empty((K,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        if sorted_route:
            flat = ids.reshape(-1)
            order = torch.sort(flat, stable=True).indices
            offsets = torch.zeros(K + 1, dtype=torch.int64, device=dev)
            offsets[1:] = torch.cumsum(torch.bincount(flat.clamp(0, K - 1), minlength=K), 0)
            N.check(lib.sn_kmeans_update_sorted(N.ptr(tokens), n_outer, n_inner, tokens.stride(0), tokens.stride(1), N.ptr(order),
                                                N.ptr(offsets), K, D, N.ptr(sums), N.ptr(counts), N.stream_ptr(dev)), "sn_kmeans_update_sorted")
        else:
            N.check(lib.sn_kmeans_update(N.ptr(tokens), n_outer, n_inner, tokens.stride(0), tokens.stride(1), N.ptr(ids), ids.stride(0),
                                         ids.stride(1), K, D, N.ptr(sums), N.ptr(counts), N.stream_ptr(dev)), "sn_kmeans_update")
    return sums, counts


def kmeans_distances(tokens, ids, centres):
    """-> float64 [n_outer * n_inner]: Euclidean distance of every token to its centre (fp64)."""
    lib = N.require_gpu()
    dev = _check_dev(tokens, ids, centres)
    if tokens.dtype != torch.float32 or tokens.stride(-1) != 1:
        tokens = tokens.to(torch.float32).contiguous()
    centres = _f32c(centres)
    n_outer, n_inner, D = tokens.shape
    assert centres.shape[1] == D and ids.dtype == torch.int64
    dist = torch.empty((n_outer * n_inner,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.sn_kmeans_distances(N.ptr(tokens), n_outer, n_inner, tokens.stride(0), tokens.stride(1), N.ptr(ids), ids.stride(0),
                                        ids.stride(1), N.ptr(centres), centres.shape[0], D, N.ptr(dist), N.stream_ptr(dev)),
                "sn_kmeans_distances")
    return dist


# ------------------------------------------------------------------------------- training loss
class _RowEntropy(torch.autograd.Function):
    """entropy over the last dimension, -sum(p * log(p + eps)) (reference schema_inference_loss.py:51-58), one
    pass forward; the backward recomputes log(p + eps) instead of keeping it and skips rows whose upstream
    gradient is zero."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, p, eps):
        lib = N.require_gpu()
        dev = _check_dev(p)
        pc = _f32c(p)
        n = pc.shape[-1]
        rows = pc.numel() // n
        ent = torch.empty(pc.shape[:-1], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.sn_row_entropy(N.ptr(pc), rows, n, float(eps), N.ptr(ent), N.stream_ptr(dev)), "sn_row_entropy")
        ctx.save_for_backward(pc)
        ctx.eps = float(eps)
        return ent

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (pc,) = ctx.saved_tensors
        lib = N.require_gpu()
        dev = pc.device
        n = pc.shape[-1]
        rows = pc.numel() // n
        gc = _f32c(g)
        grad = torch.empty_like(pc)
        with torch.cuda.device(dev):
            N.check(lib.sn_row_entropy_backward(N.ptr(pc), N.ptr(gc), rows, n, ctx.eps, N.ptr(grad), N.stream_ptr(dev)),
                    "sn_row_entropy_backward")
        return grad, None


def row_entropy(p, eps=1.0e-7):
    """-sum(p * log(p + eps), dim=-1) with autograd (CUDA tensors)."""
    return _RowEntropy.apply(p, eps)


# ------------------------------------------------------------------------------- wrapper taps
def head_mean_attention(extracted, bs):
    """extracted [bs*H, L+1, L+1] -> (attn [bs, L, L], attn_cls [bs, L])."""
    lib = N.require_gpu()
    dev = _check_dev(extracted)
    ext = _f32c(extracted)
    H = ext.shape[0] // bs
    L = ext.shape[1] - 1
    attn = torch.empty((bs, L, L), dtype=torch.float32, device=dev)
    attn_cls = torch.empty((bs, L), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.sn_head_mean_attention(N.ptr(ext), bs, H, L, N.ptr(attn), N.ptr(attn_cls), N.stream_ptr(dev)),
                "sn_head_mean_attention")
    return attn, attn_cls


# ------------------------------------------------------------------------------- S2 + S3
def _attn_view(t, L, what):
    """Accept [B, L, L] / [B, H, L, L] (any batch/head/row strides, unit column stride)."""
    if t.dtype != torch.float32 or t.stride(-1) != 1:
        t = t.to(torch.float32).contiguous()
    if t.dim() == 3:
        return t, 1, t.stride(0), t.stride(1), 0
    if t.dim() == 4:
        return t, t.shape[1], t.stride(0), t.stride(2), t.stride(1)
    raise RuntimeError(f"{what}: expected 3 or 4 dims, got {tuple(t.shape)}")


def _acls_view(t, L, what):
    if t.dtype != torch.float32 or t.stride(-1) != 1:
        t = t.to(torch.float32).contiguous()
    if t.dim() == 2:
        return t, 1, t.stride(0), 0
    if t.dim() == 3:
        return t, t.shape[1], t.stride(0), t.stride(1)
    raise RuntimeError(f"{what}: expected 2 or 3 dims, got {tuple(t.shape)}")


def instance_graph(ingredients, attn=None, attn_cls=None, *, w_v=None, w_e=None, n_pad, pad_id,
                   attn_is_logits=True, attn_cls_is_logits=True, clamp_v=None, clamp_e=None,
                   geo=None, feat_h=14, feat_w=14, dist_alpha=1.0, dist_pow=2.0, mean=True,
                   remove_self_loop=False, dicts=None, want_attr2=False, want_weighted=True,
                   attn_cls_masked_out=None, zero_padding=True, rerank=None):
    """One launch of sn_instance_graph.  Returns a dict of padded tensors:
    ids [B,n_pad] i64, n [B] i32, n_max [1] i32, and (when the inputs are given)
    v / v2 ([B,n_pad] / [B,n_pad,2]) and e / e2 ([B,n_pad,n_pad] / [...,2]).
    zero_padding=False leaves the rows / columns of e / e2 beyond an image's vertex count unwritten (two thirds of the
    padded batch): only for consumers that mask by n (gcn_adjacency_planes(..., n_valid=n)).
    rerank: the `DeferredRerank` of the S1 call that produced `ingredients` - its flagged tokens are finished inside the
    kernel's row phase (prediction configuration) or, where that does not apply, by the stand-alone re-rank first."""
    lib = N.require_gpu()
    dev = _check_dev(ingredients, attn, attn_cls, w_v, w_e, geo)
    assert ingredients.dtype == torch.int64 and ingredients.dim() == 2
    B, L = ingredients.shape
    a = N.GraphArgs()
    a.ingredients = ingredients.data_ptr()
    a.ing_stride_b, a.ing_stride_l = ingredients.stride(0), ingredients.stride(1)
    a.B, a.L = B, L
    keep = [ingredients]
    out = {}
    f32 = dict(dtype=torch.float32, device=dev)
    if attn_cls is not None:
        t, heads, sb, sh = _acls_view(attn_cls, L, "attn_cls")
        assert t.shape[0] == B and t.shape[-1] == L
        keep.append(t)
        a.attn_cls, a.acls_heads, a.acls_stride_b, a.acls_stride_h = t.data_ptr(), heads, sb, sh
        a.attn_cls_is_logits = int(attn_cls_is_logits)
        a.use_clamp_v, a.clamp_v = int(clamp_v is not None), float(clamp_v or 0.0)
        wv = _f32c(w_v.detach()).reshape(-1)
        keep.append(wv)
        a.w_v = wv.data_ptr()
        if want_weighted:
            out["v"] = torch.empty((B, n_pad), **f32)
            a.out_v = out["v"].data_ptr()
        if want_attr2:
            out["v2"] = torch.empty((B, n_pad, 2), **f32)
            a.out_v2 = out["v2"].data_ptr()
        if attn_cls_masked_out is not None:
            assert attn_cls_masked_out.is_contiguous() and tuple(attn_cls_masked_out.shape) == (B, L)
            a.attn_cls_masked = attn_cls_masked_out.data_ptr()
    if attn is not None:
        t, heads, sb, sr, sh = _attn_view(attn, L, "attn")
        assert t.shape[0] == B and t.shape[-1] == L and t.shape[-2] == L
        keep.append(t)
        a.attn, a.attn_heads = t.data_ptr(), heads
        a.attn_stride_b, a.attn_stride_r, a.attn_stride_h = sb, sr, sh
        a.attn_is_logits = int(attn_is_logits)
        a.use_clamp_e, a.clamp_e = int(clamp_e is not None), float(clamp_e or 0.0)
        we = _f32c(w_e.detach()).reshape(-1)
        keep.append(we)
        a.w_e = we.data_ptr()
        if geo is not None:
            g = _f32c(geo)
            assert tuple(g.shape) == (L, L)
            keep.append(g)
            a.geo = g.data_ptr()
        a.feat_h, a.feat_w, a.dist_alpha, a.dist_pow = feat_h, feat_w, float(dist_alpha), float(dist_pow)
        a.skip_edge_padding = int(not zero_padding)
        if want_weighted:
            out["e"] = torch.empty((B, n_pad, n_pad), **f32)
            a.out_e = out["e"].data_ptr()
        if want_attr2:
            out["e2"] = torch.empty((B, n_pad, n_pad, 2), **f32)
            a.out_e2 = out["e2"].data_ptr()
        if dicts is not None:
            keys, vals, off, ln = dicts
            keep += [keys, vals, off, ln]
            a.dict_keys, a.dict_vals, a.dict_off, a.dict_len = (keys.data_ptr(), vals.data_ptr(),
                                                                off.data_ptr(), ln.data_ptr())
    a.mean, a.remove_self_loop = int(mean), int(remove_self_loop)
    a.n_pad, a.pad_id = int(n_pad), int(pad_id)
    out["ids"] = torch.empty((B, n_pad), dtype=torch.int64, device=dev)
    out["n"] = torch.empty((B,), dtype=torch.int32, device=dev)
    out["n_max"] = torch.zeros((1,), dtype=torch.int32, device=dev)
    a.out_ids, a.out_n, a.out_n_max = out["ids"].data_ptr(), out["n"].data_ptr(), out["n_max"].data_ptr()
    rr = rerank.args_for(ingredients) if rerank is not None else None
    with torch.cuda.device(dev):
        if rr is not None:
            a.rerank = pointer(rr)
            rc = lib.sn_instance_graph(byref(a), N.stream_ptr(dev))
            if rc == 0:
                rerank.done = True
            else:                                                 # (not the prediction configuration: nothing was launched)
                a.rerank = None
                rr = None
        if rr is None:
            if rerank is not None:
                rerank.finish()
            N.check(lib.sn_instance_graph(byref(a), N.stream_ptr(dev)), "sn_instance_graph")
    del keep
    return out


# ------------------------------------------------------------------------------- init statistics
def full_vertices(ingredients, attn_cls, M, *, w_v=None, is_logits=True, clamp=None, mean=True,
                  ingredients_only=False, want_attr2=False, want_weighted=True):
    lib = N.require_gpu()
    dev = _check_dev(ingredients, attn_cls, w_v)
    B, L = ingredients.shape
    ac = _f32c(attn_cls) if attn_cls is not None else None
    wv = _f32c(w_v.detach()).reshape(-1) if w_v is not None else None
    attr2 = torch.empty((B, M, 2), dtype=torch.float32, device=dev) if want_attr2 else None
    v = torch.empty((B, M), dtype=torch.float32, device=dev) if want_weighted else None
    with torch.cuda.device(dev):
        N.check(lib.sn_full_vertices(
            N.ptr(ingredients), ingredients.stride(0), ingredients.stride(1), N.ptr(ac), B, L, int(M), int(is_logits),
            int(clamp is not None), float(clamp or 0.0), int(mean), int(ingredients_only), N.ptr(wv),
            N.ptr(attr2), N.ptr(v), N.stream_ptr(dev)), "sn_full_vertices")
    return attr2, v


def limited_edges(ingredients, attn, class_slot, label, n_max, *, w_e=None, is_logits=True, clamp=None,
                  geo=None, feat_h=14, feat_w=14, dist_alpha=1.0, dist_pow=2.0, mean=True,
                  remove_self_loop=False, want_attr2=False, want_weighted=True):
    lib = N.require_gpu()
    dev = _check_dev(ingredients, attn, class_slot, label, w_e, geo)
    B, L = ingredients.shape
    at = _f32c(attn)
    assert tuple(at.shape) == (B, L, L)
    assert class_slot.dtype == torch.int32 and class_slot.is_contiguous()
    lab = label.to(torch.int64).contiguous()
    K, Mtab = class_slot.shape
    we = _f32c(w_e.detach()).reshape(-1) if w_e is not None else None
    g = _f32c(geo) if geo is not None else None
    attr2 = torch.empty((B, n_max, n_max, 2), dtype=torch.float32, device=dev) if want_attr2 else None
    e = torch.empty((B, n_max, n_max), dtype=torch.float32, device=dev) if want_weighted else None
    with torch.cuda.device(dev):
        N.check(lib.sn_limited_edges(
            N.ptr(ingredients), ingredients.stride(0), ingredients.stride(1), N.ptr(at), B, L, int(is_logits),
            int(clamp is not None), float(clamp or 0.0), N.ptr(g), feat_h, feat_w, float(dist_alpha), float(dist_pow),
            N.ptr(class_slot), K, Mtab, N.ptr(lab), int(n_max), int(mean), int(remove_self_loop), N.ptr(we),
            N.ptr(attr2), N.ptr(e), N.stream_ptr(dev)), "sn_limited_edges")
    return attr2, e


def stats_accumulate(feat, label, class_sum, n_tracked=None):
    """class_sum[label[b]] += feat[b] in image order (in place); n_tracked[label[b]] += 1."""
    lib = N.require_gpu()
    dev = _check_dev(feat, label, class_sum, n_tracked)
    B = feat.shape[0]
    f = _f32c(feat).reshape(B, -1)
    K = class_sum.shape[0]
    assert class_sum.is_contiguous() and class_sum.dtype == torch.float32 and class_sum.numel() == K * f.shape[1]
    lab = label.to(torch.int64).contiguous()
    with torch.cuda.device(dev):
        N.check(lib.sn_stats_accumulate(N.ptr(f), N.ptr(lab), B, f.shape[1], K, N.ptr(class_sum), N.ptr(n_tracked),
                                        N.stream_ptr(dev)), "sn_stats_accumulate")
    return class_sum


# ------------------------------------------------------------------------------- atlas
def atlas_normalize(vertex_weights, edge_weights, prune_threshold=None, remove_self_loop=False):
    """-> (class_vertices [K,n], class_edges [K,n,n]); edge_weights is pruned IN PLACE."""
    lib = N.require_gpu()
    dev = _check_dev(vertex_weights, edge_weights)
    K, n = vertex_weights.shape
    assert vertex_weights.is_contiguous() and edge_weights.is_contiguous()
    cv = torch.empty((K, n), dtype=torch.float32, device=dev)
    ce = torch.empty((K, n, n), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.sn_atlas_normalize(N.ptr(vertex_weights), N.ptr(edge_weights), K, n, int(prune_threshold is not None),
                                       float(prune_threshold or 0.0), int(remove_self_loop), N.ptr(cv), N.ptr(ce),
                                       N.stream_ptr(dev)), "sn_atlas_normalize")
    return cv, ce


ENTROPY_EPS = 1.0e-7     # the eps of the loss's entropy terms (reference schema_inference_loss.py:51-58)


class _ClassEdges(torch.autograd.Function):
    """class_edges = normalised, pruned edge_weights (reference schema_net.py:152-175) with autograd: the forward pass is
    sn_atlas_normalize (one pass, pruning the parameter in place as the reference does), the backward pass
    sn_atlas_normalize_backward (one pass) - instead of seven element-wise passes over [K, n, n] forward and as many back
    (2.6 ms at the Caltech configuration's 404 MB, a sixth of a training iteration).
    entropy_eps (a number): the row entropies of class_edges (the loss's sparsity term takes a maximum over them) are a second
    output of the same pass, and the backward pass takes both upstream gradients at once - no entropy pass over class_edges,
    no [K, n, n] gradient of it that is zero outside K rows, and no pass of autograd adding that to the GCN's gradient
    (three passes over 404 MB at the Caltech configuration)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, edge_weights, vertex_weights, prune_threshold, remove_self_loop, entropy_eps=None):
        ctx.save_for_backward(edge_weights, vertex_weights)
        ctx.opts = (prune_threshold, remove_self_loop, entropy_eps)
        # (the third output: the normalised vertex weights AS THE KERNEL COMPUTED THEM - the values its pruning rule compared with the
        # threshold; whoever partitions the vertices into kept and pruned ones afterwards uses these, not a second evaluation)
        if entropy_eps is None:
            cv, ce = atlas_normalize(vertex_weights, edge_weights.detach(), prune_threshold, remove_self_loop)
            ctx.mark_non_differentiable(cv)
            return ce, cv
        lib = N.require_gpu()
        ew = edge_weights.detach()
        dev = _check_dev(vertex_weights, ew)
        K, n = vertex_weights.shape
        ce = torch.empty((K, n, n), dtype=torch.float32, device=dev)
        ent = torch.empty((K, n), dtype=torch.float32, device=dev)
        cv = torch.empty((K, n), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.sn_atlas_normalize_entropy(N.ptr(vertex_weights), N.ptr(ew), K, n, int(prune_threshold is not None),
                                                   float(prune_threshold or 0.0), int(remove_self_loop), N.ptr(cv), N.ptr(ce), N.ptr(ent),
                                                   float(entropy_eps), N.stream_ptr(dev)), "sn_atlas_normalize_entropy")
        ctx.mark_non_differentiable(cv)
        return ce, ent, cv

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_ce, *more):
        ew, vw = ctx.saved_tensors
        thr, rsl, eps = ctx.opts
        grad_ent = more[0] if eps is not None and more else None          # (outputs: ce, [ent,] cv - cv is not differentiable)
        lib = N.require_gpu()
        dev = _check_dev(ew, vw)
        K, n = vw.shape
        g = None if grad_ce is None else _f32c(grad_ce)
        ge = None if grad_ent is None else _f32c(grad_ent)
        if g is None and ge is None:
            return None, None, None, None, None
        gx = torch.empty_like(ew)
        with torch.cuda.device(dev):
            if eps is None:
                N.check(lib.sn_atlas_normalize_backward(N.ptr(vw), N.ptr(ew.detach()), N.ptr(g), K, n, int(thr is not None), float(thr or 0.0),
                                                        int(rsl), N.ptr(gx), N.stream_ptr(dev)), "sn_atlas_normalize_backward")
            else:
                N.check(lib.sn_atlas_normalize_entropy_backward(N.ptr(vw), N.ptr(ew.detach()), N.ptr(g), N.ptr(ge), float(eps), K, n,
                                                                int(thr is not None), float(thr or 0.0), int(rsl), N.ptr(gx),
                                                                N.stream_ptr(dev)), "sn_atlas_normalize_entropy_backward")
        return gx, None, None, None, None


def class_edges_autograd(edge_weights, vertex_weights, prune_threshold=None, remove_self_loop=False, with_entropy=False):
    """Differentiable class edges: edge_weights [K, n, n] (requires grad; pruned IN PLACE), vertex_weights [K, n] (detached).
    with_entropy: the result carries its row entropies (eps = ENTROPY_EPS) as `class_edges._sn_row_entropy = (eps, [K, n]
    tensor)`, a second differentiable output of the same pass: `schema_inference.loss.entropy` hands them out instead of
    reading class_edges again (and their gradient is folded into this op's one backward pass)."""
    assert edge_weights.is_contiguous() and vertex_weights.is_contiguous()
    if not with_entropy:
        ce, cv = _ClassEdges.apply(edge_weights, vertex_weights.detach(), prune_threshold, remove_self_loop)
        ce._sn_kernel_cv = cv
        return ce
    ce, ent, cv = _ClassEdges.apply(edge_weights, vertex_weights.detach(), prune_threshold, remove_self_loop, ENTROPY_EPS)
    ce._sn_row_entropy = (ENTROPY_EPS, ent)
    ce._sn_kernel_cv = cv              # (the normalised vertex weights the pruning rule of this pass saw: SchemaNet.get_atlas, compacted training)
    return ce


def atlas_adjacency_planes(vertex_weights, edge_weights, prune_threshold=None, remove_self_loop=False, want_edges=False):
    """Fused atlas route: -> (class_vertices [K,n], Planes of (E + E^T)/2 + I with E the normalised
    class edges); edge_weights is pruned IN PLACE; the [K,n,n] class_edges tensor is never written - unless
    want_edges: then it is a by-product of the same pass and a third result."""
    lib = N.require_gpu()
    dev = _check_dev(vertex_weights, edge_weights)
    K, n = vertex_weights.shape
    assert vertex_weights.is_contiguous() and edge_weights.is_contiguous()
    cv = torch.empty((K, n), dtype=torch.float32, device=dev)
    rs = torch.empty((K, n), dtype=torch.float32, device=dev)
    out = _alloc_planes(lib, dev, K, n, n)
    with torch.cuda.device(dev):
        N.check(lib.sn_atlas_prune_rowsum(N.ptr(vertex_weights), N.ptr(edge_weights), K, n, int(prune_threshold is not None),
                                          float(prune_threshold or 0.0), N.ptr(cv), N.ptr(rs), N.stream_ptr(dev)),
                "sn_atlas_prune_rowsum")
        ce = torch.empty((K, n, n), dtype=torch.float32, device=dev) if want_edges else None
        N.check(lib.sn_gcn_atlas_adjacency_planes(N.ptr(edge_weights), N.ptr(rs), K, n, int(remove_self_loop), ADJ_SCALE, N.ptr(out.hi),
                                                  N.ptr(out.lo), N.ptr(ce), N.stream_ptr(dev)), "sn_gcn_atlas_adjacency_planes")
    out.scale = const_scale(ADJ_SCALE, dev)
    return (cv, out, ce) if want_edges else (cv, out)


def atlas_adjacency_planes_compact(vertex_weights, edge_weights, prune_threshold, remove_self_loop=False, pruned_rows_are_zero=False):
    """The fused atlas route for a PRUNED atlas: -> (class_vertices [K, n], Planes of the compacted operand, perm int32 [K, n],
    n_kept int32 [K]).  perm[k] lists class k's kept vertices (class_vertices > threshold) first, in their own order, then the
    pruned ones; the operand is (E + E^T)/2 + I of the kept vertices only - an n_kept[k] x n_kept[k] corner, nothing
    produced beyond its blocks - for `gcn_gemm(..., m_extent = k_extent = rows_valid = n_kept)`.  A pruned vertex has no edge
    (schema_net.py:152-166 zeroes its row and column): it is an isolated node whose share of the class feature needs no
    product (GNN.prepare()["iso"]).  edge_weights is pruned IN PLACE like everywhere else.  No host synchronisation.
    pruned_rows_are_zero: the caller knows that an earlier call on the same versions of both parameters has run."""
    lib = N.require_gpu()
    dev = _check_dev(vertex_weights, edge_weights)
    K, n = vertex_weights.shape
    assert vertex_weights.is_contiguous() and edge_weights.is_contiguous() and prune_threshold is not None
    cv = torch.empty((K, n), dtype=torch.float32, device=dev)
    rs = torch.empty((K, n), dtype=torch.float32, device=dev)
    out = _alloc_planes(lib, dev, K, n, n)
    with torch.cuda.device(dev):
        if pruned_rows_are_zero:                     # (an earlier call on these versions of the parameters has zeroed them: not read again)
            lib.sn_atlas_skip_pruned_rows(1)
        N.check(lib.sn_atlas_prune_rowsum(N.ptr(vertex_weights), N.ptr(edge_weights), K, n, 1, float(prune_threshold), N.ptr(cv), N.ptr(rs),
                                          N.stream_ptr(dev)), "sn_atlas_prune_rowsum")
        perm = torch.empty((K, n), dtype=torch.int32, device=dev)
        n_kept = torch.empty((K,), dtype=torch.int32, device=dev)
        N.check(lib.sn_atlas_keep_perm(N.ptr(cv), K, n, float(prune_threshold), N.ptr(perm), N.ptr(n_kept), N.stream_ptr(dev)), "sn_atlas_keep_perm")
        N.check(lib.sn_gcn_atlas_adjacency_planes_compact(N.ptr(edge_weights), N.ptr(rs), K, n, int(remove_self_loop), ADJ_SCALE, N.ptr(perm),
                                                          N.ptr(n_kept), N.ptr(out.hi), N.ptr(out.lo), N.stream_ptr(dev)),
                "sn_gcn_atlas_adjacency_planes_compact")
    out.scale = const_scale(ADJ_SCALE, dev)
    return cv, out, perm, n_kept


def class_compact(perm, n_kept, nodes, ids, iso, pooled_slot=None):
    """-> (ids_c int64 [K, n], w_c [K, n], pooled_iso [K, E]): see sn_class_compact.  pooled_slot: a [K, E] view with row stride >= E
    (a slot of the [K, parts, E] partial sums `gcn_gemm(..., pooled_out=)` fills) that receives pooled_iso instead of a new tensor."""
    lib = N.require_gpu()
    dev = _check_dev(perm, n_kept, nodes, ids, iso)
    K, n = ids.shape
    nodes, iso, ids = _f32c(nodes), _f32c(iso), ids.contiguous()
    ids_c = torch.empty((K, n), dtype=torch.int64, device=dev)
    w_c = torch.empty((K, n), dtype=torch.float32, device=dev)
    pooled_iso = torch.empty((K, iso.shape[1]), dtype=torch.float32, device=dev) if pooled_slot is None else pooled_slot
    assert pooled_iso.dtype == torch.float32 and tuple(pooled_iso.shape) == (K, iso.shape[1]) and pooled_iso.stride(1) == 1 and pooled_iso.device == dev
    with torch.cuda.device(dev):
        N.check(lib.sn_class_compact(N.ptr(perm), N.ptr(n_kept), N.ptr(nodes), N.ptr(ids), N.ptr(iso), K, n, iso.shape[1], iso.shape[0],
                                     N.ptr(ids_c), N.ptr(w_c), N.ptr(pooled_iso), pooled_iso.stride(0) if K > 1 else iso.shape[1],
                                     N.stream_ptr(dev)), "sn_class_compact")
    return ids_c, w_c, pooled_iso


# ------------------------------------------------------------------------------- S4
def gcn_adjacency(edges):
    lib = N.require_gpu()
    dev = _check_dev(edges)
    e = _f32c(edges)
    G, n, _ = e.shape
    adj = torch.empty_like(e)
    with torch.cuda.device(dev):
        N.check(lib.sn_gcn_adjacency(N.ptr(e), G, n, N.ptr(adj), N.stream_ptr(dev)), "sn_gcn_adjacency")
    return adj


def _dp(t):
    return None if t is None else t.data_ptr()


ADJ_SCALE = 1024.0     # static scale of adjacency planes: (E + E^T)/2 + I of normalised graphs is <= 2 + |w_e|_1; room up to 63 (caller-supplied edges beyond that saturate at the fp16 maximum in the producer instead of becoming inf: csrc/sn_gcn.hip, saturate_f16_range)
_PLANE_TOP = 8192.0    # a scaled operand's largest magnitude lands in [2^13, 2^14)
_const_scales = {}


def const_scale(value, dev):
    """device scalar (fp32 [1]) holding a compile-time scale, one per (value, device)"""
    key = (float(value), str(dev))
    if key not in _const_scales:
        _const_scales[key] = torch.full((1,), float(value), dtype=torch.float32, device=dev)
    return _const_scales[key]


def pow2_scale(bound):
    """bound: tensor (its largest magnitude is taken) or 0-dim / [1] tensor holding a bound on |x| -> device scalar fp32 [1],
    the power of two s with s * bound in [2^13, 2^14) (clamped to 2^-60 .. 2^60; a zero / non-finite bound gives 1): what
    split-fp16 planes of x are scaled by (csrc/sn_gcn.hip, "What hi + lo holds").  No host synchronisation."""
    b = bound.detach()
    if b.is_cuda and b.dtype == torch.float32 and b.numel() > 1 and b.is_contiguous():
        # a whole operand: one read of it and one finishing launch (csrc/sn_train.hip) instead of abs, amax and nine scalar launches
        lib = N.require_gpu()
        partial = torch.empty(lib.sn_pow2_scale_blocks(b.numel()), dtype=torch.int32, device=b.device)
        s = torch.empty(1, dtype=torch.float32, device=b.device)
        with torch.cuda.device(b.device):
            N.check(lib.sn_pow2_scale(N.ptr(b), b.numel(), _PLANE_TOP, N.ptr(partial), N.ptr(s), None, N.stream_ptr(b.device)), "sn_pow2_scale")
        return s
    b = b.abs().amax().to(torch.float32).reshape(1)
    s = torch.exp2(torch.floor(torch.log2(_PLANE_TOP / b)).clamp(-60.0, 60.0))
    return torch.where(torch.isfinite(s) & (b > 0), s, torch.ones_like(s))


def sym_half_(s):
    """s [G, n, n] fp32 contiguous (CUDA) <- (s + s^T) / 2 per graph, in place (sn_sym_half_inplace)."""
    lib = N.require_gpu()
    dev = _check_dev(s)
    assert s.dtype == torch.float32 and s.is_contiguous() and s.dim() == 3 and s.shape[1] == s.shape[2]
    with torch.cuda.device(dev):
        N.check(lib.sn_sym_half_inplace(N.ptr(s), s.shape[0], s.shape[1], N.stream_ptr(dev)), "sn_sym_half_inplace")
    return s


def normalize_sum_rows_(x, min_val=0.0, zero_diagonal=False):
    """In place on x [..., n] fp32 contiguous (CUDA): clamp_min(min_val), divide every row by its sum, NaN -> 0, and (x = [K, n, n],
    zero_diagonal) the diagonal to 0 afterwards: `SchemaNet.normalize()` on one parameter as one pass (reference
    schema_net.py:133-142, graph/utils.py:7-13)."""
    lib = N.require_gpu()
    dev = _check_dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() >= 1
    n = x.shape[-1]
    assert not zero_diagonal or (x.dim() == 3 and x.shape[1] == n)
    if x.numel():
        with torch.cuda.device(dev):
            N.check(lib.sn_normalize_sum_rows(N.ptr(x), x.numel() // n, n, float(min_val), n if zero_diagonal else 0, N.stream_ptr(dev)),
                    "sn_normalize_sum_rows")
    return x


class Planes:
    """A [batches, rows, k] fp32 operand as two blocked fp16 planes (include/schemanet_hip.h, S4):
    hi/lo are flat [batches, plane_elems] tensors; `rows`/`k` are the logical extent, `kpad` the
    k of the blocks (multiple of 16); `scale`: None or a device scalar (fp32 [1], a power of two) - the planes hold
    x * scale."""

    __slots__ = ("hi", "lo", "batches", "rows", "k", "kpad", "scale", "pending", "grad_sum", "compact")

    def __init__(self, hi, lo, batches, rows, k, scale=None):
        self.hi, self.lo, self.batches, self.rows, self.k, self.scale = hi, lo, batches, rows, k, scale
        self.pending, self.grad_sum = 0, None       # training: products whose backward is still to come / the sum of their dY . X^T
        self.compact = None                         # training with compacted class graphs: (perm int32 [G, n], n_kept int32 [G]) of an adjacency operand
        self.kpad = (k + 15) // 16 * 16

    @property
    def stride(self):
        return self.hi.shape[1] if self.batches > 1 else 0

    def to_dense(self):
        """[batches, rows, k] fp32 reconstruction hi + lo (tests / debugging)."""
        rb, kb = (self.rows + 31) // 32, self.kpad // 16
        def un(p):
            x = p.float().view(self.batches, rb, kb, 2, 32, 8)           # [g, rb, kb, h, r, e]
            x = x.permute(0, 1, 4, 2, 3, 5).reshape(self.batches, rb * 32, kb * 16)
            return x
        full = un(self.hi) + un(self.lo)
        if self.scale is not None:
            full = full / self.scale
        return full[:, :self.rows, :self.k], full


def _alloc_planes(lib, dev, batches, rows, k):
    elems = lib.sn_gcn_plane_elems(int(rows), int(k))
    hi = torch.empty((batches, elems), dtype=torch.float16, device=dev)
    return Planes(hi, torch.empty_like(hi), batches, rows, k)


def gcn_adjacency_planes(edges, extent=None, n_valid=None, per_graph=False):
    """(E + E^T)/2 + I of [G, n, n] edges as blocked fp16 hi/lo planes.  extent: optional int32 [1]
    device tensor (largest vertex count of the batch): blocks beyond it are not produced.  n_valid: optional int32 [G]
    vertex counts: edges outside a graph's own corner count as zero and are not read.  per_graph (with n_valid): every graph's
    own count is its extent - for gcn_gemm(..., m_extent=n_valid, k_extent=n_valid) only."""
    lib = N.require_gpu()
    dev = _check_dev(edges)
    e = _f32c(edges)
    G, n, _ = e.shape
    out = _alloc_planes(lib, dev, G, n, n)
    with torch.cuda.device(dev):
        if n_valid is not None and per_graph:
            assert n_valid.dtype == torch.int32 and n_valid.numel() == G and n_valid.device == dev and n_valid.is_contiguous()
            N.check(lib.sn_gcn_adjacency_planes_per_graph(N.ptr(e), G, n, N.ptr(n_valid), ADJ_SCALE, N.ptr(out.hi), N.ptr(out.lo), N.stream_ptr(dev)),
                    "sn_gcn_adjacency_planes_per_graph")
        elif n_valid is not None:
            assert n_valid.dtype == torch.int32 and n_valid.numel() == G and n_valid.device == dev
            N.check(lib.sn_gcn_adjacency_planes_masked(N.ptr(e), G, n, N.ptr(n_valid), N.ptr(extent), ADJ_SCALE, N.ptr(out.hi), N.ptr(out.lo),
                                                       N.stream_ptr(dev)), "sn_gcn_adjacency_planes_masked")
        else:
            N.check(lib.sn_gcn_adjacency_planes(N.ptr(e), G, n, N.ptr(extent), ADJ_SCALE, N.ptr(out.hi), N.ptr(out.lo), N.stream_ptr(dev)),
                    "sn_gcn_adjacency_planes")
    out.scale = const_scale(ADJ_SCALE, dev)
    return out


def gcn_adjacency_planes_compact(edges, perm, n_kept):
    """(E + E^T)/2 + I of [G, n, n] edges through a vertex permutation: vertex a of graph g's operand is vertex perm[g, a] of `edges`
    (the kept vertices first: atlas_keep_perm), only the n_kept[g] x n_kept[g] corner (+ identity) is produced - the operand of the
    training route with compacted class graphs (`planes.compact` = (perm, n_kept); products take n_kept as their extents)."""
    lib = N.require_gpu()
    dev = _check_dev(edges, perm, n_kept)
    e = _f32c(edges)
    G, n, _ = e.shape
    assert perm.dtype == torch.int32 and perm.is_contiguous() and tuple(perm.shape) == (G, n) and n_kept.dtype == torch.int32 and n_kept.numel() == G
    out = _alloc_planes(lib, dev, G, n, n)
    with torch.cuda.device(dev):
        N.check(lib.sn_gcn_adjacency_planes_compact(N.ptr(e), G, n, N.ptr(perm), N.ptr(n_kept), ADJ_SCALE, N.ptr(out.hi), N.ptr(out.lo), N.stream_ptr(dev)),
                "sn_gcn_adjacency_planes_compact")
    out.scale = const_scale(ADJ_SCALE, dev)
    out.compact = (perm, n_kept.contiguous())
    return out


def atlas_keep_perm(class_vertices, prune_threshold):
    """-> (perm int32 [K, n]: the vertices of every class with class_vertices > threshold first, in their own order, then the others;
    n_kept int32 [K]).  No host synchronisation."""
    lib = N.require_gpu()
    dev = _check_dev(class_vertices)
    cv = _f32c(class_vertices.detach())
    K, n = cv.shape
    perm = torch.empty((K, n), dtype=torch.int32, device=dev)
    n_kept = torch.empty((K,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.sn_atlas_keep_perm(N.ptr(cv), K, n, float(prune_threshold), N.ptr(perm), N.ptr(n_kept), N.stream_ptr(dev)), "sn_atlas_keep_perm")
    return perm, n_kept


def sym_scatter_corner(corner, perm, n_kept):
    """corner [G, n, n] (rows < n_kept[g] written, compacted vertex order) -> the symmetrised edge gradient [G, n, n] in the stored vertex
    order, zero where a pruned vertex is involved (sn_sym_scatter_corner)."""
    lib = N.require_gpu()
    dev = _check_dev(corner, perm, n_kept)
    assert corner.dtype == torch.float32 and corner.is_contiguous() and corner.dim() == 3 and corner.shape[1] == corner.shape[2]
    out = torch.empty_like(corner)
    with torch.cuda.device(dev):
        N.check(lib.sn_sym_scatter_corner(N.ptr(corner), N.ptr(perm), N.ptr(n_kept), corner.shape[0], corner.shape[1], N.ptr(out), N.stream_ptr(dev)),
                "sn_sym_scatter_corner")
    return out


def gcn_gather_planes(table, ids, extent=None, scale=None):
    """Zt[g, f, j] = table[ids[g, j], f] as blocked planes of a [G, E, n] operand (times the device scalar `scale`)."""
    lib = N.require_gpu()
    dev = _check_dev(table, ids)
    t = _f32c(table.detach())
    ids = ids.contiguous()
    assert ids.dtype == torch.int64
    G, n = ids.shape
    rows, E = t.shape
    out = _alloc_planes(lib, dev, G, E, n)
    with torch.cuda.device(dev):
        N.check(lib.sn_gcn_gather_planes(N.ptr(t), rows, N.ptr(ids), G, n, E, N.ptr(extent), N.ptr(scale), N.ptr(out.hi), N.ptr(out.lo),
                                         N.stream_ptr(dev)), "sn_gcn_gather_planes")
    out.scale = scale
    return out


def split_planes(x, scale="auto", transpose=False, node_extents=None):
    """fp32 [rows, k] or [batches, rows, k] -> blocked hi/lo planes with hi + lo ~= x * scale (22 significant bits down to
    2^-17 of the largest magnitude).  scale: "auto" = pow2_scale(x) (one reduction over x, no host synchronisation), a
    device scalar from pow2_scale(bound) when a bound on |x| is known without reading x, or None (= 1: only for operands
    known to lie in 2^-3 .. 6e4).  transpose: the planes of x^T ([k, rows] per batch entry), read from x as it is."""
    lib = N.require_gpu()
    dev = _check_dev(x)
    xc = _f32c(x.detach())
    if xc.dim() == 2:
        xc = xc[None]
    B_, rows, k = xc.shape
    if isinstance(scale, str):
        scale = pow2_scale(xc) if xc.numel() else None
    out = _alloc_planes(lib, dev, B_, k, rows) if transpose else _alloc_planes(lib, dev, B_, rows, k)
    with torch.cuda.device(dev):
        if node_extents is not None:
            # x = [G, nodes, features] of graphs with their own node counts (compacted class graphs): blocks of pad nodes only are not produced
            assert node_extents.dtype == torch.int32 and node_extents.numel() == B_ and node_extents.is_contiguous() and node_extents.device == dev
            N.check(lib.sn_split_planes_nodes(N.ptr(xc), B_, rows, k, N.ptr(scale), N.ptr(node_extents), int(bool(transpose)), N.ptr(out.hi), N.ptr(out.lo),
                                              N.stream_ptr(dev)), "sn_split_planes_nodes")
            out.scale = scale
            return out
        fn = lib.sn_split_planes_transposed if transpose else lib.sn_split_planes
        N.check(fn(N.ptr(xc), B_, rows, k, k, rows * k, N.ptr(scale), N.ptr(out.hi), N.ptr(out.lo), N.stream_ptr(dev)),
                "sn_split_planes_transposed" if transpose else "sn_split_planes")
    out.scale = scale
    return out


def table_planes(table, scale=None):
    """fp32 [rows, E] (E a multiple of 256) -> (hi, lo) row-major fp16 planes [rows + 1, E] with hi + lo ~= table * scale (the same split as
    split_planes) and a zero last row: the gathered-B operand of gcn_gemm (pass the same device scalar as `b_scale`)."""
    t = _f32c(table.detach())
    if scale is not None:
        t = t * scale
    hi = t.to(torch.float16)
    lo = (t - hi.to(torch.float32)).to(torch.float16)
    z = torch.zeros((1, t.shape[1]), dtype=torch.float16, device=t.device)
    return torch.cat([hi, z]).contiguous(), torch.cat([lo, z]).contiguous()


def next_layer_weight_planes(weight):
    """Planes of a [256, 256] Linear weight with its columns in the order the GEMM epilogue holds the features
    (sn_gemm_args.next_w_*): the operand of `gcn_gemm(..., next_w=)`."""
    assert tuple(weight.shape) == (256, 256)
    kappa = torch.arange(256, device=weight.device)
    col = (4 * (kappa >> 7) + (kappa & 3)) * 32 + ((kappa >> 2) & 31)
    return split_planes(weight.detach()[:, col].contiguous())            # (scale: from the weight's largest magnitude)


def gcn_gemm(a, b, batches, bias=None, layernorm=None, relu=False, rows_valid=None,
             want_c=False, want_planes=0, pool_w=None, m_extent=None, k_extent=None, zero_c=False, b_table=None, next_w=None,
             out_scale=None, h_scale=None, accumulate_into=None, pooled_out=None):
    """C[g] = A[g] . Bt[g]^T on split-fp16 planes (sn_gcn_gemm): A = Planes [*, m, k], Bt = Planes [*, n, k].

    layernorm: (gamma, beta, eps) or None.  want_planes: 0, or the k extent of the result planes
    (>= n; the extra columns are zero).  pool_w [batches, m] -> "pooled" [batches, ceil(m/128), n]:
    per-row-tile partial sums of sum_m pool_w[m] C[m, :] (add them up, or hand them to pool_fc).
    zero_c: the fp32 result starts as zeros (row tiles beyond m_extent are never written).
    b_table = (hi, lo, ids): B is gathered inside the kernel, Bt[g, f, j] = table[ids[g, j], f] (hi, lo = table_planes(table),
    ids int64 [batches, n_ids]); `b` is then None.  256 features with the LayerNorm epilogue, any multiple of 256 without.
    next_w = next_layer_weight_planes(W): the epilogue result H [m, 256] is not stored; "planes" is Zt = W . H^T as a
    [256, want_planes >= m] operand (the next GraphConv's Linear, fused: no H round trip, one launch less).
    Scales: the operands' `.scale` (b_table: an optional 4th entry) are divided out of the accumulators; out_scale
    (device scalar from pow2_scale(bound on the result)) is what the output planes are multiplied by and carry as their
    `.scale`; h_scale (with next_w): the bound-derived scale of the epilogue's H fragments.
    accumulate_into: fp32 [batches, m, n] contiguous - the plain product is ADDED to it (and it is returned as "c").
    pooled_out: fp32 [batches, parts >= ceil(m / 128), n] contiguous that receives the partial sums in its first slots (the
    further ones are the caller's: class_compact's pooled_iso) and is returned as "pooled".
    Returns dict(c=fp32 [batches, m, n], planes=Planes, pooled=...)."""
    lib = N.require_gpu()
    args = N.GemmArgs()
    b_scale = None
    if b_table is not None:
        if len(b_table) == 4:
            b_scale = b_table[3]
        t_hi, t_lo, ids = b_table[:3]
        dev = _check_dev(a.hi, a.lo, t_hi, t_lo, ids)
        assert b is None and a.batches in (1, batches) and ids.dtype == torch.int64 and ids.is_contiguous() and ids.shape[0] == batches
        assert t_hi.dtype == torch.float16 and t_hi.is_contiguous() and t_lo.is_contiguous() and t_hi.shape == t_lo.shape and t_hi.shape[1] % 256 == 0
        m, n = a.rows, t_hi.shape[1]
        args.b_table_hi, args.b_table_lo, args.b_ids = _dp(t_hi), _dp(t_lo), _dp(ids)
        args.b_ids_stride, args.b_ids_n, args.b_table_rows = ids.shape[1], ids.shape[1], t_hi.shape[0] - 1
    else:
        dev = _check_dev(a.hi, a.lo, b.hi, b.lo)
        assert a.kpad == b.kpad and a.batches in (1, batches) and b.batches in (1, batches)
        m, n = a.rows, b.rows
        args.b_hi, args.b_lo, args.b_batch_stride = _dp(b.hi), _dp(b.lo), b.stride
        b_scale = b.scale
    args.a_hi, args.a_lo, args.a_batch_stride = _dp(a.hi), _dp(a.lo), a.stride
    args.a_scale, args.b_scale, args.out_scale = _dp(a.scale), _dp(b_scale), _dp(out_scale)
    for sc in (a.scale, b_scale, out_scale, h_scale):
        assert sc is None or (sc.dtype == torch.float32 and sc.numel() == 1 and sc.device == dev)
    args.m, args.n, args.k, args.batches = int(m), int(n), int(a.kpad), int(batches)
    out = {}
    keep = []
    if accumulate_into is not None:
        c = accumulate_into
        assert c.dtype == torch.float32 and c.is_contiguous() and tuple(c.shape) == (batches, m, n) and c.device == dev
        assert not want_planes and pool_w is None and bias is None and layernorm is None and not relu and rows_valid is None
        args.c, args.c_batch_stride, args.ldc, args.accumulate = _dp(c), m * n, n, 1
        out["c"] = c
    elif want_c:
        # zero_c with a row mask: the kernel defines every element itself (rows_valid zeroes the pad rows of the tiles it multiplies, the
        # workgroups of skipped tiles write their zeros) - no clearing pass over the result
        in_kernel = bool(zero_c and m_extent is not None and rows_valid is not None and rows_valid is m_extent and m_extent.numel() == batches)
        c = (torch.zeros if (zero_c and not in_kernel) else torch.empty)((batches, m, n), dtype=torch.float32, device=dev)
        args.c, args.c_batch_stride, args.ldc = _dp(c), m * n, n
        args.zero_skipped = int(in_kernel)
        out["c"] = c
    if next_w is not None:
        assert want_planes and not want_c and pool_w is None and n == 256 and layernorm is not None and next_w.rows == 256 and next_w.kpad == 256
        args.next_w_hi, args.next_w_lo = _dp(next_w.hi), _dp(next_w.lo)
        args.next_w_scale, args.next_h_scale = _dp(next_w.scale), _dp(h_scale)
        cp = _alloc_planes(lib, dev, batches, 256, int(want_planes))
        cp.scale = out_scale
        args.c_hi, args.c_lo, args.cp_batch_stride, args.cp_cols = _dp(cp.hi), _dp(cp.lo), cp.hi.shape[1], cp.kpad
        out["planes"] = cp
    elif want_planes:
        cp = _alloc_planes(lib, dev, batches, m, int(want_planes))
        cp.scale = out_scale
        args.c_hi, args.c_lo, args.cp_batch_stride, args.cp_cols = _dp(cp.hi), _dp(cp.lo), cp.hi.shape[1], cp.kpad
        out["planes"] = cp
    if bias is not None:
        bt = _f32c(bias.detach()); keep.append(bt)
        args.bias = _dp(bt)
    if layernorm is not None:
        g_, b_, eps = layernorm
        g_, b_ = _f32c(g_.detach()), _f32c(b_.detach()); keep += [g_, b_]
        args.gamma, args.beta, args.eps, args.layernorm = _dp(g_), _dp(b_), float(eps), 1
    args.relu = int(bool(relu))
    if rows_valid is not None:
        assert rows_valid.dtype == torch.int32
        args.rows_valid = _dp(rows_valid)
    if pool_w is not None:
        pw = _f32c(pool_w); keep.append(pw)
        assert pw.shape == (batches, m)
        if pooled_out is None:
            pooled = torch.empty((batches, (m + 127) // 128, n), dtype=torch.float32, device=dev)
        else:
            pooled = pooled_out
            assert pooled.dtype == torch.float32 and pooled.is_contiguous() and pooled.device == dev and pooled.dim() == 3
            assert pooled.shape[0] == batches and pooled.shape[1] >= (m + 127) // 128 and pooled.shape[2] == n
            args.pooled_parts = pooled.shape[1]
        args.pool_w, args.pool_w_stride, args.pooled = _dp(pw), m, _dp(pooled)
        out["pooled"] = pooled
    for name, t in (("m_extent", m_extent), ("k_extent", k_extent)):
        if t is not None:
            assert t.dtype == torch.int32 and t.device == dev and t.numel() in (1, batches)
            setattr(args, name, _dp(t))
            if t.numel() == batches and batches > 1:         # one extent per graph (compacted class graphs)
                args.extent_stride = 1
    assert not args.extent_stride or all(t is None or t.numel() == batches for t in (m_extent, k_extent))
    with torch.cuda.device(dev):
        N.check(lib.sn_gcn_gemm(byref(args), N.stream_ptr(dev)), "sn_gcn_gemm")
    return out


class _SymAdjMatmul(torch.autograd.Function):
    """Y = adj @ X for a SYMMETRIC adj [G, n, n] (the GCN operand (E + E^T)/2 + I, reference gnn.py:27-30) on the matrix
    cores with autograd: the training-side counterpart of the inference products.  All three
    products are split-fp16 MFMA GEMMs (sn_gcn_gemm, fp32-GEMM accuracy):
        forward    Y     = adj . X                 A = adj planes (shared by the layers),  Bt = planes of X^T
        backward   dX    = adj . dY                (adj symmetric: adj^T = adj)
                   d adj = dY . X^T                A = planes of dY [G, n, E], Bt = planes of X [G, n, E]: no transposes
    `adj_planes` = split_planes(adj) is passed in so that the layers of one forward pass share it."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, adj, x, adj_planes):
        G, n, _ = adj.shape
        xt = split_planes(x.detach(), transpose=True)
        y = gcn_gemm(adj_planes, xt, G, want_c=True)["c"]
        ctx.save_for_backward(x)
        ctx.adj_planes = adj_planes
        ctx.adj_like = torch.empty(adj.shape, dtype=adj.dtype, device="meta")
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        ap = ctx.adj_planes
        G = x.shape[0]
        # Gradients are small (1e-3 .. 1e-8) and fp16 planes have a short exponent: split_planes scales every operand by
        # the power of two that puts its largest magnitude at 2^13 (a device scalar, no host synchronisation), the
        # product divides the scales out again.
        dy = _f32c(dy)
        if dy.numel() == 0:                                  # G == 0 or n == 0: nothing to multiply
            return (torch.zeros(ctx.adj_like.shape, dtype=ctx.adj_like.dtype, device=x.device) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(x) if ctx.needs_input_grad[1] else None, None)
        d_adj = d_x = None
        if ctx.needs_input_grad[1]:
            d_x = gcn_gemm(ap, split_planes(dy, transpose=True), G, want_c=True)["c"]
        if ctx.needs_input_grad[0]:
            d_adj = gcn_gemm(split_planes(dy), split_planes(x.detach()), G, want_c=True)["c"]
        return d_adj, d_x, None


def _edge_grad_of(s_, adj_planes):
    """S = sum of the layers' dY . X^T -> dE = (S + S^T) / 2; for a compacted operand: its corner symmetrised and scattered back to
    the stored vertex order (zero where a pruned vertex is involved)"""
    if adj_planes.compact is not None:
        return sym_scatter_corner(s_, *adj_planes.compact)
    return sym_half_(s_)


class _EdgesAdjMatmul(torch.autograd.Function):
    """Y = ((E + E^T)/2 + I) @ X straight from the edge tensor E [G, n, n] (reference gnn.py:27-31): the adjacency only ever
    exists as the fp16 planes `adj_planes` = gcn_adjacency_planes(E) (one pass over E; no dense adjacency, no identity
    matrix, no transpose copy), shared by the layers of one forward pass and by their backward passes.
        backward   dX = adj . dY,      dE = (S + S^T) / 2  with  S = dY . X^T   (the chain rule through the symmetrisation)"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, edges, x, adj_planes, sum_edge_grads=False):
        G = edges.shape[0]
        ext = adj_planes.compact[1] if adj_planes.compact is not None else None      # compacted class graphs: one extent per graph
        xt = split_planes(x.detach(), transpose=True, node_extents=ext)
        y = gcn_gemm(adj_planes, xt, G, want_c=True, zero_c=ext is not None, m_extent=ext, k_extent=ext, rows_valid=ext)["c"]
        ctx.save_for_backward(x)
        ctx.adj_planes = adj_planes
        ctx.x_scale = xt.scale                               # (the backward's planes of x: no second reduction over it)
        ctx.adj_like = torch.empty(edges.shape, dtype=edges.dtype, device="meta")
        ctx.counted = bool(sum_edge_grads) and bool(ctx.needs_input_grad[0]) and os.environ.get("SN_GCN_SUM_DE", "1") != "0"
        if ctx.counted:
            adj_planes.pending += 1
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        ap = ctx.adj_planes
        G = x.shape[0]
        dy = _f32c(dy)
        if dy.numel() == 0:
            return (torch.zeros(ctx.adj_like.shape, dtype=ctx.adj_like.dtype, device=x.device) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(x) if ctx.needs_input_grad[1] else None, None, None)
        d_e = d_x = None
        dy_scale = pow2_scale(dy)                            # one reduction over dy for both of its plane forms
        ext = ap.compact[1] if ap.compact is not None else None
        if ctx.needs_input_grad[1]:
            d_x = gcn_gemm(ap, split_planes(dy, scale=dy_scale, transpose=True, node_extents=ext), G, want_c=True, zero_c=ext is not None, m_extent=ext, k_extent=ext,
                           rows_valid=ext)["c"]
        if ctx.needs_input_grad[0]:
            dyp, xp = split_planes(dy, scale=dy_scale, node_extents=ext), split_planes(x.detach(), scale=ctx.x_scale, node_extents=ext)
            if not ctx.counted:
                s_ = gcn_gemm(dyp, xp, G, want_c=True, m_extent=ext)["c"]
                return _edge_grad_of(s_, ap), d_x, None, None
            # The layers of one forward pass share `ap` and the edge tensor: their S = dY . X^T are summed in ONE buffer (the
            # product's epilogue adds) and symmetrised once, by the layer whose backward runs last; the others hand autograd
            # no gradient for the edges (= zero).  Per layer this was a strided add, a scale and autograd's accumulation:
            # three more passes over a [G, n, n] tensor (404 MB at config [4]'s real size).
            if ap.grad_sum is None:
                ap.grad_sum = gcn_gemm(dyp, xp, G, want_c=True, m_extent=ext)["c"]
            else:
                gcn_gemm(dyp, xp, G, accumulate_into=ap.grad_sum, m_extent=ext)
            ap.pending -= 1
            if ap.pending == 0:
                d_e, ap.grad_sum = _edge_grad_of(ap.grad_sum, ap), None
        return d_e, d_x, None, None


def edges_adj_matmul(edges, x, adj_planes=None, sum_edge_grads=False):
    """edges [G, n, n] fp32, x [G, n, E] fp32 (CUDA, E a multiple of 16) -> ((edges + edges^T)/2 + I) @ x, differentiable in both.
    sum_edge_grads: the caller applies several of these products IN SEQUENCE to the same `edges` with the same `adj_planes` (the
    layers of a GNN: every one of them is on the path to the loss if the last is) - their edge gradients are then summed in one
    buffer and handed to autograd once, by the product whose backward runs last."""
    if adj_planes is None:
        adj_planes = gcn_adjacency_planes(edges.detach())
    return _EdgesAdjMatmul.apply(edges, x, adj_planes, sum_edge_grads)


def sym_adj_matmul(adj, x, adj_planes=None):
    """adj [G, n, n] symmetric fp32, x [G, n, E] fp32 (CUDA, E a multiple of 16) -> adj @ x, differentiable in both."""
    if adj_planes is None:
        adj_planes = split_planes(adj.detach())
    return _SymAdjMatmul.apply(adj, x, adj_planes)


def _plane_batches(p, g0, g1):
    """the planes of the batch entries g0 .. g1 - 1 (views)"""
    return Planes(p.hi[g0:g1], p.lo[g0:g1], g1 - g0, p.rows, p.k, p.scale)


_DW_TEMP_BYTES = 256 << 20     # largest [group, out, in] fp32 temporary of a per-graph weight gradient


def _weight_grad_per_graph(dyt, xt, extents=None):
    """sum_g dY[g]^T X[g] from the planes of dY^T [G, out, n] and X^T [G, in, n]: G products of inner length n and one sum over G
    (one [out, in] product of inner length G n would be (out / 128) x (in / 256) workgroups walking 100 k rows each); graphs in
    groups when the [G, out, in] temporary would pass _DW_TEMP_BYTES (config [3]: 1000 class graphs of width 1024 = 4.2 GB).
    extents (int32 [G]): the inner length of graph g (rows beyond it are zero rows of dY)."""
    G, out_f, in_f = dyt.batches, dyt.rows, xt.rows
    group = max(1, min(G, _DW_TEMP_BYTES // max(1, 4 * out_f * in_f)))
    if group >= G:
        return gcn_gemm(dyt, xt, G, want_c=True, k_extent=extents)["c"].sum(dim=0)
    dw = None
    for g0 in range(0, G, group):
        g1 = min(G, g0 + group)
        part = gcn_gemm(_plane_batches(dyt, g0, g1), _plane_batches(xt, g0, g1), g1 - g0, want_c=True,
                        k_extent=None if extents is None else extents[g0:g1].contiguous())["c"].sum(dim=0)
        dw = part if dw is None else dw.add_(part)
    return dw


class _LinearMfma(torch.autograd.Function):
    """y = x W^T (+ b) on x [G, n, in] (or [rows, in]) with all three products - forward, dX = dY W, dW = sum_g dY[g]^T X[g] - as
    split-fp16 MFMA GEMMs (sn_gcn_gemm: fp32-GEMM accuracy): the training side's Linear layers (reference gnn.py:31,
    `self.linear(torch.bmm(adj, feat))`) without a library GEMM.
        forward   A = planes of X [G, n, in],        Bt = planes of W [out, in] (shared by the graphs), bias in the epilogue
        dX        A = planes of dY [G, n, out],      Bt = planes of W^T [in, out]
        dW        A = planes of dY^T [G, out, n],    Bt = planes of X^T [G, in, n]: per graph, summed over G"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias, extents=None):
        x3 = _f32c(x.detach())
        x3 = x3[None] if x3.dim() == 2 else x3
        G = x3.shape[0]
        xp, wp = split_planes(x3, node_extents=extents), split_planes(weight.detach())
        # extents (int32 [G], graphs of [G, n, in] only): rows >= extents[g] of graph g are pad rows - neither multiplied nor written (zeros)
        y = gcn_gemm(xp, wp, G, bias=bias, want_c=True, zero_c=extents is not None, m_extent=extents, rows_valid=extents)["c"]
        ctx.save_for_backward(x, weight)
        ctx.scales = (xp.scale, wp.scale)
        ctx.has_bias = bias is not None
        ctx.extents = extents
        return y[0] if x.dim() == 2 else y

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        x_scale, w_scale = ctx.scales
        dx = dw = db = None
        dy3, x3 = _f32c(dy), _f32c(x.detach())
        if x.dim() == 2:
            dy3, x3 = dy3[None], x3[None]
        G = x3.shape[0]
        if dy3.numel() == 0:
            return (torch.zeros_like(x) if ctx.needs_input_grad[0] else None, torch.zeros_like(weight) if ctx.needs_input_grad[1] else None,
                    torch.zeros(weight.shape[0], dtype=weight.dtype, device=weight.device) if ctx.has_bias and ctx.needs_input_grad[2] else None, None)
        ext = ctx.extents
        dy_scale = pow2_scale(dy3)                              # one reduction over dY for both of its plane forms
        if ctx.needs_input_grad[0]:
            dx = gcn_gemm(split_planes(dy3, scale=dy_scale, node_extents=ext), split_planes(weight.detach(), scale=w_scale, transpose=True), G, want_c=True,
                          zero_c=ext is not None, m_extent=ext, rows_valid=ext)["c"]
            dx = dx[0] if x.dim() == 2 else dx
        if ctx.needs_input_grad[1]:
            dw = _weight_grad_per_graph(split_planes(dy3, scale=dy_scale, transpose=True, node_extents=ext),
                                        split_planes(x3, scale=x_scale, transpose=True, node_extents=ext), ext)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy3.sum(dim=(0, 1))
        return dx, dw, db, None


def linear_mfma(x, weight, bias=None, extents=None):
    """x [G, n, in] or [rows, in] fp32 (CUDA; in and out multiples of 16), weight [out, in], bias [out] or None -> x W^T + b,
    differentiable in all three, every product on the matrix cores (no library GEMM).  extents (int32 [G], 3-D x only): the rows
    of graph g from extents[g] on are pad rows whose gradient is zero (compacted class graphs): they come back as zeros."""
    assert extents is None or (x.dim() == 3 and extents.dtype == torch.int32 and extents.numel() == x.shape[0])
    return _LinearMfma.apply(x, weight, bias, extents)


class _GatherAdjMatmul(torch.autograd.Function):
    """Y = ((E + E^T)/2 + I) @ table[ids] + bias from the edge tensor E [G, n, n], a table [rows, F] and ids [G, n]: layer 1 of
    the GCN in training with its Linear folded into the embedding table (`table` = embedding.weight @ W1^T, a [rows, F] product
    instead of the [G n, F] one of reference gnn.py:31 - the same re-association the inference route makes, gnn.py `prepare`).
    The gathered operand only ever exists as fp16 planes (sn_gcn_gather_planes); no [G, n, F] fp32 embedding is written.
        backward   dTable = scatter-add by ids of adj . dY   (a cached sort of the ids, or the library's embedding backward)
                   dE = (S + S^T)/2, S = dY . table[ids]^T   (summed over the layers that share `adj_planes`, as _EdgesAdjMatmul)
                   dBias = sum of dY over graphs and vertices"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, edges, table, ids, bias, adj_planes, sort, padding_idx, sum_edge_grads):
        G = edges.shape[0]
        tab = _f32c(table.detach())
        t_scale = pow2_scale(tab)
        zt = gcn_gather_planes(tab, ids, scale=t_scale)
        ext = adj_planes.compact[1] if adj_planes.compact is not None else None      # compacted class graphs (ids in the operand's vertex order)
        y = gcn_gemm(adj_planes, zt, G, bias=bias, want_c=True, zero_c=ext is not None, m_extent=ext, k_extent=ext, rows_valid=ext)["c"]
        ctx.save_for_backward(tab, ids, *(sort if sort is not None else ()))
        ctx.adj_planes, ctx.t_scale, ctx.has_bias = adj_planes, t_scale, bias is not None
        ctx.pad = int(padding_idx) if padding_idx is not None else -1
        ctx.adj_like = torch.empty(edges.shape, dtype=edges.dtype, device="meta")
        ctx.counted = bool(sum_edge_grads) and bool(ctx.needs_input_grad[0]) and os.environ.get("SN_GCN_SUM_DE", "1") != "0"
        if ctx.counted:
            adj_planes.pending += 1
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        tab, ids, *sort = ctx.saved_tensors
        ap = ctx.adj_planes
        G, n = ids.shape
        rows, F = tab.shape
        dy = _f32c(dy)
        d_e = d_tab = d_b = None
        if dy.numel() == 0:
            return (torch.zeros(ctx.adj_like.shape, dtype=ctx.adj_like.dtype, device=tab.device) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(tab) if ctx.needs_input_grad[1] else None, None,
                    torch.zeros(F, dtype=tab.dtype, device=tab.device) if ctx.has_bias and ctx.needs_input_grad[3] else None, None, None, None, None)
        dy_scale = pow2_scale(dy)
        ext = ap.compact[1] if ap.compact is not None else None
        if ctx.needs_input_grad[1]:
            dz = gcn_gemm(ap, split_planes(dy, scale=dy_scale, transpose=True, node_extents=ext), G, want_c=True, zero_c=ext is not None, m_extent=ext, k_extent=ext,
                          rows_valid=ext)["c"].reshape(-1, F)
            if sort:
                lib = N.require_gpu()
                d_tab = torch.empty((rows, F), dtype=torch.float32, device=dz.device)
                with torch.cuda.device(dz.device):
                    N.check(lib.sn_embedding_grad_sorted(N.ptr(dz), N.ptr(sort[0]), N.ptr(sort[1]), rows, F, ctx.pad, N.ptr(d_tab), N.stream_ptr(dz.device)),
                            "sn_embedding_grad_sorted")
            elif ids.numel() <= 32768 and F <= 1024 and os.environ.get("SN_EMBED_SCAN", "1") != "0":
                # ids that are new in every iteration (the instance graphs' words): one launch, no sort (sn_embedding_grad_scan)
                lib = N.require_gpu()
                d_tab = torch.empty((rows, F), dtype=torch.float32, device=dz.device)
                with torch.cuda.device(dz.device):
                    N.check(lib.sn_embedding_grad_scan(N.ptr(dz), N.ptr(ids), ids.numel(), rows, F, ctx.pad, N.ptr(d_tab), N.stream_ptr(dz.device)),
                            "sn_embedding_grad_scan")
            else:
                d_tab = torch.ops.aten.embedding_dense_backward(dz, ids, rows, ctx.pad, False)
        if ctx.needs_input_grad[0]:
            dyp = split_planes(dy, scale=dy_scale, node_extents=ext)
            zp = split_planes(torch.nn.functional.embedding(ids, tab), scale=ctx.t_scale, node_extents=ext)
            if not ctx.counted:
                d_e = _edge_grad_of(gcn_gemm(dyp, zp, G, want_c=True, m_extent=ext)["c"], ap)
            else:
                if ap.grad_sum is None:
                    ap.grad_sum = gcn_gemm(dyp, zp, G, want_c=True, m_extent=ext)["c"]
                else:
                    gcn_gemm(dyp, zp, G, accumulate_into=ap.grad_sum, m_extent=ext)
                ap.pending -= 1
                if ap.pending == 0:
                    d_e, ap.grad_sum = _edge_grad_of(ap.grad_sum, ap), None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            d_b = dy.sum(dim=(0, 1))
        return d_e, d_tab, None, d_b, None, None, None, None


def gather_adj_matmul(edges, table, ids, bias=None, adj_planes=None, sort=None, padding_idx=None, sum_edge_grads=False):
    """edges [G, n, n], table [rows, F] (F a multiple of 16), ids int64 [G, n] with values in [0, rows) -> ((edges + edges^T)/2 + I) @
    table[ids] + bias [G, n, F], differentiable in edges, table and bias.  sort = sorted_ids_of(ids, rows) when the caller keeps it
    (ids that stay the same over the iterations); padding_idx: that row of the table gets no gradient (nn.Embedding's rule).
    sum_edge_grads: as edges_adj_matmul."""
    if adj_planes is None:
        adj_planes = gcn_adjacency_planes(edges.detach())
    return _GatherAdjMatmul.apply(edges, table, ids.contiguous(), bias, adj_planes, sort, padding_idx, sum_edge_grads)


def pool_fc(pooled_sum, divisor, weight, bias, weight_t=None):
    """fc(pooled / divisor) with pooled = pooled_sum [G, E] or the sum over dim 1 of [G, parts, E]
    (the per-row-tile partial sums of gcn_gemm); divisor: int32 [1] device tensor or a number.
    weight_t: weight.t().contiguous() kept by the caller (a weight-only operand): the faster kernel form."""
    lib = N.require_gpu()
    dev = _check_dev(pooled_sum, weight, bias)
    p = _f32c(pooled_sum)
    transposed = weight_t is not None and p.shape[-1] <= 2048
    w = _f32c(weight_t.detach()) if transposed else _f32c(weight.detach())
    E_out = weight.shape[0]
    assert not transposed or tuple(w.shape) == (weight.shape[1], E_out)
    b = None if bias is None else _f32c(bias.detach())
    if p.dim() == 2:
        p = p[:, None, :]
    G, parts, E = p.shape
    out = torch.empty((G, E_out), dtype=torch.float32, device=dev)
    if torch.is_tensor(divisor):
        assert divisor.dtype == torch.int32 and divisor.device == dev
        ddev, dhost = N.ptr(divisor), 0.0
    else:
        ddev, dhost = None, float(divisor)
    with torch.cuda.device(dev):
        fn = lib.sn_pool_fc_t if transposed else lib.sn_pool_fc
        N.check(fn(N.ptr(p), G, parts, E, ddev, dhost, N.ptr(w), N.ptr(b), E_out, N.ptr(out), N.stream_ptr(dev)),
                "sn_pool_fc_t" if transposed else "sn_pool_fc")
    return out


class _EmbeddingSortedGrad(torch.autograd.Function):
    """table[ids] whose backward takes the sort of `ids` from the caller (sorted_ids_of: done once for an index tensor that
    stays the same from iteration to iteration) instead of sorting in every backward pass."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, weight, ids, order, seg, padding_idx):
        ctx.save_for_backward(order, seg)
        ctx.opts = (tuple(weight.shape), int(padding_idx) if padding_idx is not None else -1, tuple(ids.shape))
        return torch.nn.functional.embedding(ids, weight.detach())

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        order, seg = ctx.saved_tensors
        (rows, E), pad, _ = ctx.opts
        lib = N.require_gpu()
        dyc = _f32c(dy).reshape(-1, E)
        dev = _check_dev(dyc, order, seg)
        grad = torch.empty((rows, E), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.sn_embedding_grad_sorted(N.ptr(dyc), N.ptr(order), N.ptr(seg), rows, E, pad, N.ptr(grad), N.stream_ptr(dev)),
                    "sn_embedding_grad_sorted")
        return grad, None, None, None, None


def sorted_ids_of(ids, rows):
    """-> (order int64 [N], seg int64 [rows + 1]) of an index tensor with values in [0, rows): see sn_embedding_grad_sorted"""
    flat = ids.reshape(-1)
    srt, order = torch.sort(flat, stable=True)
    seg = torch.searchsorted(srt, torch.arange(rows + 1, device=ids.device, dtype=flat.dtype))
    return order.contiguous(), seg.contiguous()


def embedding_sorted(weight, ids, order, seg, padding_idx=None):
    """weight[ids] (fp32 CUDA table [rows, E], E % 4 == 0), differentiable in `weight`; (order, seg) = sorted_ids_of(ids, rows)."""
    assert weight.dtype == torch.float32 and weight.shape[1] % 4 == 0 and order.dtype == torch.int64 and seg.numel() == weight.shape[0] + 1
    return _EmbeddingSortedGrad.apply(weight, ids, order, seg, padding_idx)


class _MaskLayerNormAct(torch.autograd.Function):
    """y = act(LayerNorm(x with the rows >= n_valid[g] set to 0)) on x [G, n, E] (reference gnn.py:43-46) with autograd: one HIP
    pass forward (the values of mask_layernorm_act_), one back over (x, dy) that recomputes the row statistics and yields dx,
    d gamma, d beta (csrc/sn_train.hip) - for the library's three passes forward and four back."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, gamma, beta, eps, n_valid, relu):
        lib = N.require_gpu()
        xc, gc, bc = _f32c(x.detach()), _f32c(gamma.detach()), _f32c(beta.detach())
        dev = _check_dev(xc, gc, bc, n_valid)
        G, n, E = xc.shape
        y = torch.empty_like(xc)
        with torch.cuda.device(dev):
            N.check(lib.sn_mask_layernorm_act_forward(N.ptr(xc), N.ptr(y), G, n, E, N.ptr(n_valid), N.ptr(gc), N.ptr(bc), float(eps), int(relu),
                                                      N.stream_ptr(dev)), "sn_mask_layernorm_act_forward")
        ctx.save_for_backward(xc, gc, bc, n_valid)
        ctx.opts = (float(eps), int(relu))
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        xc, gc, bc, n_valid = ctx.saved_tensors
        eps, relu = ctx.opts
        lib = N.require_gpu()
        dev = xc.device
        G, n, E = xc.shape
        dyc = _f32c(dy)
        dx = torch.empty_like(xc)
        dgb = torch.empty((2, E), dtype=torch.float32, device=dev)
        partial = torch.empty((max(1, lib.sn_ln_act_blocks(G * n)), 2, E), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.sn_mask_layernorm_act_backward(N.ptr(xc), N.ptr(dyc), G, n, E, N.ptr(n_valid), N.ptr(gc), N.ptr(bc), eps, relu,
                                                       N.ptr(dx), N.ptr(partial), N.ptr(dgb), N.stream_ptr(dev)), "sn_mask_layernorm_act_backward")
        return dx, dgb[0], dgb[1], None, None, None


def mask_layernorm_act(x, gamma, beta, eps, n_valid=None, relu=True):
    """Differentiable, out of place: x [G, n, E] fp32 (CUDA, E <= 1024), n_valid int32 [G] or None."""
    assert x.dim() == 3 and (n_valid is None or (n_valid.dtype == torch.int32 and n_valid.numel() == x.shape[0]))
    return _MaskLayerNormAct.apply(x, gamma, beta, eps, n_valid, relu)


def mask_layernorm_act_(x, gamma, beta, eps, n_valid=None, relu=True):
    """In place on x [G, n, E]."""
    lib = N.require_gpu()
    dev = _check_dev(x, gamma, beta, n_valid)
    assert x.is_contiguous() and x.dtype == torch.float32
    G, n, E = x.shape
    with torch.cuda.device(dev):
        N.check(lib.sn_mask_layernorm_act(N.ptr(x), G, n, E, N.ptr(n_valid), N.ptr(_f32c(gamma.detach())),
                                          N.ptr(_f32c(beta.detach())), float(eps), int(relu), N.stream_ptr(dev)),
                "sn_mask_layernorm_act")
    return x


def weighted_pool(feat, nodes, divisor_dev=None):
    lib = N.require_gpu()
    dev = _check_dev(feat, nodes, divisor_dev)
    f, w = _f32c(feat), _f32c(nodes)
    G, n, E = f.shape
    out = torch.empty((G, E), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.sn_weighted_pool(N.ptr(f), N.ptr(w), G, n, E, N.ptr(divisor_dev), N.ptr(out), N.stream_ptr(dev)),
                "sn_weighted_pool")
    return out


class _WeightedPool(torch.autograd.Function):
    """weighted_pool with autograd (training): one pass forward, one back (sn_weighted_pool_backward) - the library's
    `(feat * nodes[..., None]).sum(1) / div` is a product, a reduction and, backwards, two products and a reduction over [G, n, E]."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, feat, nodes, divisor_dev):
        f, w = _f32c(feat.detach()), _f32c(nodes.detach())
        ctx.save_for_backward(f, w, divisor_dev)
        return weighted_pool(f, w, divisor_dev)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        f, w, div = ctx.saved_tensors
        lib = N.require_gpu()
        dev = f.device
        G, n, E = f.shape
        gc = _f32c(g)
        d_feat = torch.empty_like(f)
        d_nodes = torch.empty_like(w)
        with torch.cuda.device(dev):
            N.check(lib.sn_weighted_pool_backward(N.ptr(f), N.ptr(w), N.ptr(gc), G, n, E, N.ptr(div), N.ptr(d_feat), N.ptr(d_nodes),
                                                  N.stream_ptr(dev)), "sn_weighted_pool_backward")
        return d_feat, d_nodes, None


def weighted_pool_autograd(feat, nodes, divisor_dev=None):
    """Differentiable in feat [G, n, E] (E % 4 == 0) and nodes [G, n]; divisor_dev: int32 [1] device tensor or None (= n)."""
    return _WeightedPool.apply(feat, nodes, divisor_dev)


def layernorm_weighted_pool(x, gamma, beta, eps, nodes, n_valid=None, relu=True, divisor_dev=None):
    """weighted_pool(mask_layernorm_act_(x), nodes, divisor_dev) in one pass over x [G, n, E]; x is left as it is."""
    lib = N.require_gpu()
    dev = _check_dev(x, gamma, beta, nodes, n_valid, divisor_dev)
    assert x.is_contiguous() and x.dtype == torch.float32
    G, n, E = x.shape
    out = torch.empty((G, E), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.sn_layernorm_weighted_pool(N.ptr(x), N.ptr(_f32c(nodes)), G, n, E, N.ptr(n_valid), N.ptr(_f32c(gamma.detach())),
                                               N.ptr(_f32c(beta.detach())), float(eps), int(relu), N.ptr(divisor_dev), N.ptr(out),
                                               N.stream_ptr(dev)), "sn_layernorm_weighted_pool")
    return out


def layernorm_split_planes(x, gamma, beta, eps, n_valid=None, relu=True, scale=None):
    """split_planes(mask_layernorm_act_(x), scale) in one pass over x [G, n, E] (E % 16 == 0); x is left as it is.
    `scale`: a device scalar (pow2_scale of a bound on the LayerNorm output) or None."""
    lib = N.require_gpu()
    dev = _check_dev(x, gamma, beta, n_valid)
    assert x.is_contiguous() and x.dtype == torch.float32
    G, n, E = x.shape
    out = _alloc_planes(lib, dev, G, n, E)
    with torch.cuda.device(dev):
        N.check(lib.sn_layernorm_split_planes(N.ptr(x), G, n, E, N.ptr(n_valid), N.ptr(_f32c(gamma.detach())), N.ptr(_f32c(beta.detach())),
                                              float(eps), int(relu), N.ptr(scale), N.ptr(out.hi), N.ptr(out.lo), N.stream_ptr(dev)),
                "sn_layernorm_split_planes")
    out.scale = scale
    return out


class _WeighAttributes(torch.autograd.Function):
    """attr2 [..., 2] x w [2, 1] -> [...] with the gradient of the two weights (attr2 itself carries none: the attributes are counts,
    geometry and attention values of the HIP forward): sn_weigh_attributes / _backward, one launch forward and two backward (round 6;
    as torch ops - two products, an add, four selects of the weights and their backward nodes - it was ~30 launches of an iteration)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, attr2, w):
        lib = N.require_gpu()
        dev = _check_dev(attr2, w)
        a = _f32c(attr2.detach())
        wc = _f32c(w.detach()).reshape(2)
        out = torch.empty(a.shape[:-1], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.sn_weigh_attributes(N.ptr(a), out.numel(), N.ptr(wc), N.ptr(out), N.stream_ptr(dev)), "sn_weigh_attributes")
        ctx.save_for_backward(a)
        ctx.w_shape, ctx.w_dtype = tuple(w.shape), w.dtype
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None
        lib = N.require_gpu()
        dev = a.device
        gc = _f32c(g)
        dw = torch.zeros(2, dtype=torch.float32, device=dev) if gc.numel() == 0 else torch.empty(2, dtype=torch.float32, device=dev)
        if gc.numel():
            partial = torch.empty(lib.sn_weigh_blocks(gc.numel()) * 2, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                N.check(lib.sn_weigh_attributes_backward(N.ptr(a), N.ptr(gc), gc.numel(), N.ptr(partial), N.ptr(dw), N.stream_ptr(dev)),
                        "sn_weigh_attributes_backward")
        return None, dw.reshape(ctx.w_shape).to(ctx.w_dtype)


def weigh_attributes(attr2, w):
    """attr2 [..., 2] (count / geometry, attention) and the attribute weights w [2, 1] -> attr2 @ w, squeezed, visible to autograd
    (the reference runs this matmul inside its C++, large_scale_feat_to_e.cpp:141-147).  The library runs a [B n n, 2] x [2, 1]
    product on 16 x 16 tiles - 2.9 ms for the 2.4 M edge cells of a 64-image batch - and its weight gradient as a second such product;
    here one HIP pass each way (`SN_WEIGH_FUSED=0`: two multiplies and an add as torch ops, rounds 3-5: the same values forward)."""
    if attr2.is_cuda and not attr2.requires_grad and attr2.dtype == torch.float32 and w.numel() == 2 and attr2.shape[-1] == 2 \
            and os.environ.get("SN_WEIGH_FUSED", "1") != "0":
        return _WeighAttributes.apply(attr2, w)
    if attr2.is_cuda:
        return attr2[..., 0] * w[0, 0] + attr2[..., 1] * w[1, 0]
    return (attr2 @ w).squeeze(-1)


class _RectifyLinear(torch.autograd.Function):
    """x if x > a else a - 1 + 1 / (1 + a - x) on a CUDA tensor (the loss's rectified sparsity terms: scalars) as one launch, its
    derivative kept from the forward: one multiply back."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, a):
        lib = N.require_gpu()
        dev = _check_dev(x)
        xc = _f32c(x.detach())
        out, deriv = torch.empty_like(xc), torch.empty_like(xc)
        with torch.cuda.device(dev):
            N.check(lib.sn_rectify_linear(N.ptr(xc), xc.numel(), float(a), N.ptr(out), N.ptr(deriv), N.stream_ptr(dev)), "sn_rectify_linear")
        ctx.save_for_backward(deriv)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (deriv,) = ctx.saved_tensors
        return g * deriv, None


def rectify_linear(x, a):
    """the select form of the reference's rectify_linear (schema_inference_loss.py:61-67) for a CUDA fp32 tensor"""
    return _RectifyLinear.apply(x, a)


SIMILARITY = {"inner_product": 0, "cosine": 1, "euclidean": 2}


def match_scores(feat_inst, feat_kg, similarity="inner_product", votes=None):
    """pred [B, K]; votes (f32 [K + 1], optional): the per-class argmax counts of these scores are added to it in the same
    launch (== class_votes_(pred, votes))."""
    lib = N.require_gpu()
    dev = _check_dev(feat_inst, feat_kg, votes)
    a, b = _f32c(feat_inst), _f32c(feat_kg)
    B, E = a.shape
    K = b.shape[0]
    pred = torch.empty((B, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if votes is not None:
            assert votes.dtype == torch.float32 and votes.is_contiguous() and votes.numel() == K + 1
            N.check(lib.sn_match_scores_votes(N.ptr(a), N.ptr(b), B, K, E, SIMILARITY[similarity], N.ptr(pred), N.ptr(votes),
                                              N.stream_ptr(dev)), "sn_match_scores_votes")
        else:
            N.check(lib.sn_match_scores(N.ptr(a), N.ptr(b), B, K, E, SIMILARITY[similarity], N.ptr(pred), N.stream_ptr(dev)),
                    "sn_match_scores")
    return pred


def class_votes_(pred, votes):
    """votes [K + 1] f32 += per-class argmax counts of pred [B, K] (and the image count)."""
    lib = N.require_gpu()
    dev = _check_dev(pred, votes)
    p = _f32c(pred)
    B, K = p.shape
    assert votes.dtype == torch.float32 and votes.is_contiguous() and votes.numel() == K + 1
    with torch.cuda.device(dev):
        N.check(lib.sn_class_votes(N.ptr(p), B, K, N.ptr(votes), N.stream_ptr(dev)), "sn_class_votes")
    return votes
