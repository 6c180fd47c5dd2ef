"""numpy restatement of the reference's Python-level tensor ops on the schema-inference path.
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Each function cites the reference lines it follows (paths relative to /root/reference).
All arithmetic is float32 unless `dtype=np.float64` is passed (used for error analysis).
"""
import numpy as np

from . import cabi

F32 = np.float32
_FMAX = np.finfo(np.float32).max


# ----------------------------------------------------------------------------- small helpers
def nan_to_num0(x):
    """torch.nan_to_num(x, 0): nan->0, +-inf -> +-float32 max."""
    return np.nan_to_num(x, nan=0.0, posinf=_FMAX, neginf=-_FMAX)


def clamp_softmax(x, clamp, nan_to_zero, dtype=F32):
    """schema_net.py:295-297 (vertex: nan_to_zero=True) and :334-336 (edge: False).
    Returns (masked logits as the in-place masked_fill_ leaves them, softmax)."""
    x = np.array(x, dtype=dtype, copy=True)
    if clamp is not None:
        x[x < dtype(clamp)] = -np.inf
    with np.errstate(invalid="ignore", over="ignore"):
        mx = x.max(axis=-1, keepdims=True)
        e = np.exp(x - mx)
        sm = e / e.sum(axis=-1, keepdims=True)
    if nan_to_zero:
        sm = nan_to_num0(sm)
    return x, sm.astype(dtype)


def pair_wise_point_sim(h, w, alpha=1.0, pow=2.0):
    """schema_inference/graph/utils.py:55-81.  geo[p,q] = 1/(1 + |grid_p-grid_q|_pow/alpha),
    p = r*w + c.  Integer coordinates: every intermediate is exact up to one correctly
    rounded sqrt / div, so this is bit-reproducible."""
    r, c = np.meshgrid(np.arange(h, dtype=F32), np.arange(w, dtype=F32), indexing="ij")
    p = np.stack((r.reshape(-1), c.reshape(-1)), axis=1)
    d = np.abs(p[:, None, :] - p[None, :, :])
    if pow == 2:
        dist = np.sqrt((d * d).sum(-1, dtype=F32), dtype=F32)
    else:
        dist = np.power(np.power(d, F32(pow)).sum(-1, dtype=F32), F32(1.0 / pow)).astype(F32)
    dist = dist / F32(alpha)
    return (F32(1) / (F32(1) + dist)).astype(F32)


def normalize_sum_(x, axis):
    """graph/utils.py:7-13"""
    with np.errstate(invalid="ignore", divide="ignore"):
        return nan_to_num0(x / x.sum(axis=axis, keepdims=True)).astype(x.dtype)


def normalize_max_(x, axis):
    """graph/utils.py:16-22"""
    with np.errstate(invalid="ignore", divide="ignore"):
        return nan_to_num0(x / x.max(axis=axis, keepdims=True)).astype(x.dtype)


# ----------------------------------------------------------------------------- S1
def discretize(mid_feat, codebook, activate=True):
    """DiscretizationJitWrapper.forward (scripts/save_backbone_jit.py:127-131) =
    Adapter.adapt -> Discretization.encode (discretization.py:58-70) -> Adapter.reconstruct.
    mid_feat [L+1, bs, D] seq-first.  -> (seq [L+1, bs, D], ingredients i64 [L, bs])."""
    mid_feat = np.asarray(mid_feat, F32)
    cls, seq = mid_feat[:1], mid_feat[1:]
    n, bs, d = seq.shape
    idx = cabi.assign_words(seq.reshape(n * bs, d), codebook)
    if activate:
        seq = np.asarray(codebook, F32)[idx].reshape(n, bs, d)
    return np.concatenate((cls, seq), axis=0), idx.reshape(n, bs)


# ----------------------------------------------------------------------------- wrapper (a3)
def wrapper_attention(extracted, bs):
    """ingredient_model_wrapper.py:58-68: head-mean of raw logits, then slicing.
    extracted [bs*H, L+1, L+1] -> attn [bs, L, L], attn_cls [bs, L]."""
    e = np.asarray(extracted, F32)
    a = e.reshape(bs, -1, e.shape[1], e.shape[2]).mean(axis=1, dtype=F32)
    return np.ascontiguousarray(a[:, 1:, 1:]), np.ascontiguousarray(a[:, 0, 1:])


# ----------------------------------------------------------------------------- S2 / S3
def instance_graph(ing, attn, attn_cls, w_v, w_e, clamp_v=-1.0, clamp_e=-1.0,
                   feat_h=14, feat_w=14, alpha=1.0, pow=2.0, remove_self_loop=False):
    """SchemaNet.forward (schema_net.py:377-399) on logits.  Returns dict of python lists like
    the reference plus the masked logits (the in-place side effect on the caller's tensors)."""
    ing = np.asarray(ing, np.int64)
    m_cls, sm_cls = clamp_softmax(attn_cls, clamp_v, nan_to_zero=True)
    ids, a2, wts, num_v = cabi.instance_v(ing, sm_cls, w_v, mean=True)
    splits = np.cumsum(num_v)[:-1]
    inst_ids = np.split(ids, splits)
    inst_v = np.split(wts, splits)
    inst_v2 = np.split(a2, splits)
    m_attn, sm_attn = clamp_softmax(attn, clamp_e, nan_to_zero=False)
    geo = pair_wise_point_sim(feat_h, feat_w, alpha, pow)
    dicts = [{int(v): k for k, v in enumerate(i.tolist())} for i in inst_ids]
    e2, e = cabi.instance_e(ing, sm_attn, geo, dicts, w_e, mean=True, remove_self_loop=remove_self_loop)
    return {
        "instance_ingredients": inst_ids, "instance_vertices": inst_v, "instance_edges": e,
        "instance_vertices_attr2": inst_v2, "instance_edges_attr2": e2,
        "attn_cls_masked": m_cls, "attn_masked": m_attn, "num_v": num_v,
    }


# ----------------------------------------------------------------------------- init path
def full_vertices(ing, attn_cls, n_vertices, w_v, clamp_v=-1.0):
    """SchemaNet.feat_to_full_vertices (schema_net.py:188-207): no nan_to_num after softmax."""
    _, sm = clamp_softmax(attn_cls, clamp_v, nan_to_zero=False)
    attr = cabi.v_attr(np.asarray(ing, np.int64), sm, n_vertices, mean=True)
    attr = normalize_max_(attr, axis=1)
    return (attr @ np.asarray(w_v, F32).reshape(2, 1))[..., 0].astype(F32)


def limited_edges(ing, attn, label, class_slot, n_max, w_e, clamp_e=-1.0, feat_h=14, feat_w=14,
                  alpha=1.0, pow=2.0, remove_self_loop=False):
    """SchemaNet.feat_to_limited_edges (schema_net.py:222-254)."""
    _, sm = clamp_softmax(attn, clamp_e, nan_to_zero=False)
    geo = pair_wise_point_sim(feat_h, feat_w, alpha, pow)
    e = cabi.feat_to_e(np.asarray(ing, np.int64), sm, geo, class_slot, label, n_max, mean=True)
    e = normalize_sum_(e, axis=2)
    if remove_self_loop:
        i = np.arange(n_max)
        e[:, i, i, :] = 0
    return (e @ np.asarray(w_e, F32).reshape(2, 1))[..., 0].astype(F32)


def init_class_vertices(ing, attn_cls, label, n_classes, n_vertices, w_v, clamp_v=-1.0):
    """scripts/init_schema_net.py:43-65 over ONE pass of (ing, attn_cls, label).
    Returns (class_vertices [K, M] normalised, raw sums, n_tracked)."""
    v = full_vertices(ing, attn_cls, n_vertices, w_v, clamp_v)
    sums = np.zeros((n_classes, n_vertices), F32)
    n = np.zeros(n_classes, F32)
    for k, vi in zip(np.asarray(label).tolist(), v):
        sums[k] += vi
        n[k] += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        cv = sums / n[:, None]
        cv = cv / cv.sum(axis=-1, keepdims=True)
    return cv.astype(F32), sums, n


def init_graph(ing, attn, label, class_slot, n_classes, n_max, w_e, **kw):
    """scripts/init_schema_net.py:19-40 before graph.normalize(): per-class mean of the
    limited edges.  Returns (edge_weights [K,n_max,n_max], raw sums, n_tracked)."""
    e = limited_edges(ing, attn, label, class_slot, n_max, w_e, **kw)
    sums = np.zeros((n_classes, n_max, n_max), F32)
    n = np.zeros(n_classes, F32)
    for k, ei in zip(np.asarray(label).tolist(), e):
        sums[k] += ei
        n[k] += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        ew = sums / n[:, None, None]
    return ew.astype(F32), sums, n


# ----------------------------------------------------------------------------- atlas (a9)
def normalize_sum_clamp(x, min_val=0.0):
    """graph/utils.py:25-52 (detach only matters for autograd)."""
    x = np.maximum(x, x.dtype.type(min_val))
    with np.errstate(invalid="ignore", divide="ignore"):
        return nan_to_num0(x / x.sum(axis=-1, keepdims=True)).astype(x.dtype)


def get_atlas(vertex_weights, edge_weights, prune_node_threshold=0.001, remove_self_loop=False):
    """SchemaNet.get_atlas (schema_net.py:144-184).  Returns (class_vertices, class_edges,
    edge_weights_after) -- the last is the Parameter after the in-place masked_fill_ (:164)."""
    vw = np.asarray(vertex_weights, F32)
    ew = np.array(edge_weights, F32, copy=True)
    cv = normalize_sum_clamp(vw, 1.0e-5)
    if prune_node_threshold is not None:
        mask = (cv > F32(prune_node_threshold)).astype(F32)
        mask = mask[:, :, None] * mask[:, None, :]
        ew[mask == 0] = 0
        e = ew * mask
    else:
        e = ew
    ce = normalize_sum_clamp(e, 0.0)
    if remove_self_loop:
        i = np.arange(ce.shape[1])
        ce[:, i, i] = 0
    return cv, ce, ew


# ----------------------------------------------------------------------------- S4 (a10, a11)
def _layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True, dtype=x.dtype)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=x.dtype)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * g + b


def gnn_forward(params, nodes, edges, ids, feat_mask=None, dtype=F32):
    """GNN.forward / Layer.forward / GraphConv.forward (gnn.py:78-98, 41-46, 20-31), relu.
    params: state-dict-like mapping with the reference's key names under 'gnn.'."""
    P = {k: np.asarray(v, dtype) for k, v in params.items()}
    feat = P["gnn.embedding.weight"][np.asarray(ids, np.int64)]
    edges = np.asarray(edges, dtype)
    n_layers = len([k for k in P if k.endswith("g_conv.linear.weight")])
    eye = np.eye(edges.shape[-1], dtype=dtype)
    adj = (edges + np.swapaxes(edges, 1, 2)) / dtype(2) + eye
    for i in range(n_layers):
        feat = adj @ feat
        feat = feat @ P[f"gnn.layers.{i}.g_conv.linear.weight"].T + P[f"gnn.layers.{i}.g_conv.linear.bias"]
        if feat_mask is not None:
            feat = np.where(np.asarray(feat_mask, bool)[..., None], dtype(0), feat)
        feat = _layer_norm(feat, P[f"gnn.layers.{i}.norm.weight"], P[f"gnn.layers.{i}.norm.bias"])
        feat = np.maximum(feat, dtype(0))
    feat = feat * np.asarray(nodes, dtype)[..., None]
    feat = feat.mean(axis=1, dtype=dtype)  # divides by the PADDED length (gnn.py:96)
    return feat @ P["gnn.fc.weight"].T + P["gnn.fc.bias"]


def matcher_forward(params, inst_ids, inst_v, inst_e, class_vertices, class_edges, class_ingredients,
                    num_codes, similarity="inner_product", dtype=F32):
    """Matcher.forward (match.py:33-76): pad ragged graphs to max n_i, GNN on both sides,
    similarity.  Returns pred [B, K]."""
    bs = len(inst_ids)
    sizes = [len(x) for x in inst_ids]
    n = max(sizes)
    ids = np.full((bs, n), num_codes, np.int64)
    v = np.zeros((bs, n), dtype)
    e = np.zeros((bs, n, n), dtype)
    mask = np.zeros((bs, n), bool)
    for i, s in enumerate(sizes):
        ids[i, :s] = inst_ids[i]
        v[i, :s] = inst_v[i]
        e[i, :s, :s] = inst_e[i]
        mask[i, s:] = True
    f_inst = gnn_forward(params, v, e, ids, mask, dtype)
    f_kg = gnn_forward(params, class_vertices, class_edges, class_ingredients, None, dtype)
    a, b = f_inst[:, None, :], f_kg[None, :, :]
    if similarity == "inner_product":
        return (a * b).sum(-1)
    if similarity == "cosine":
        na = np.maximum(np.linalg.norm(a, axis=-1), 1e-8)
        nb = np.maximum(np.linalg.norm(b, axis=-1), 1e-8)
        return ((a * b).sum(-1) / (na * nb) + 1) / 2
    if similarity == "euclidean":
        return 1 / (1 + np.linalg.norm(a - b, axis=-1))
    raise KeyError(similarity)


# ------------------------------------------------------------------------------- codebook extraction
def kmeans_lloyd(obs, guess, thresh=1e-5, max_iter=10_000):
    """SciPy's `_kmeans(obs, guess, thresh)` (scipy/cluster/vq.py), float32 path, restated: exact nearest
    centre (first index on ties), fp32 member sums in observation order / count, centres without members
    dropped, stop when the mean distance changes by <= thresh.  -> (book f32 [k', D], avg distance, iterations).
    Reference call site: scripts/extract_ingredients.py:33-36."""
    obs = np.ascontiguousarray(obs, F32)
    book = np.ascontiguousarray(guess, F32)
    prev, diff, it = [np.inf], np.inf, 0
    while diff > thresh and it < max_iter:
        ids = cabi.assign_words(obs, book)
        dist = cabi.kmeans_distances(obs, ids, book)
        prev = (prev + [float(dist.sum() / dist.size)])[-2:]
        sums, counts = cabi.kmeans_update(obs, ids, book.shape[0])
        has = counts > 0
        book = np.ascontiguousarray(sums[has] / counts[has].astype(F32)[:, None])
        diff = abs(prev[0] - prev[1])
        it += 1
    return book, prev[1], it


# ------------------------------------------------------------------------------- training loss
def row_entropy(p, eps=1.0e-7):
    """-sum(p * log(p + eps), -1) in float64 (reference schema_inference_loss.py:51-58)."""
    p = np.asarray(p, np.float64)
    return -(p * np.log(p + eps)).sum(-1)


def row_entropy_grad(p, g, eps=1.0e-7):
    p = np.asarray(p, np.float64)
    return -np.asarray(g, np.float64)[..., None] * (np.log(p + eps) + p / (p + eps))


def schema_inference_loss(pred, label, class_vertices, class_edges, a_vertex=3.0, a_edge=3.0):
    """reference SchemaInferenceLoss.forward (schema_inference_loss.py:21-47), float64."""
    pred = np.asarray(pred, np.float64)
    z = pred - pred.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    cls = -logp[np.arange(len(label)), np.asarray(label)].mean()
    ev = row_entropy(class_vertices).max(0)
    ee = row_entropy(class_edges).max(1).mean()
    rect = lambda x, a: x if x > a else a - 1 + 1.0 / (1 + a - x)  # noqa: E731
    return {"cls": cls, "entropy_vertex": ev, "entropy_edge": ee, "re_entropy_vertex": rect(ev, a_vertex), "re_entropy_edge": rect(ee, a_edge)}
