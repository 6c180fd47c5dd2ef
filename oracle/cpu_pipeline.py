"""CPU baseline pipeline: the reference's forward, op for op, on the host cores.
TEST / BENCH INFRASTRUCTURE ONLY (bench.py `cpu_baseline` leg and tests).

Mirrors SchemaNetPredictor.forward (reference schema_inference/graph/__init__.py:37-57) with the
same torch ops the reference calls (torch.cdist + argmin, softmax, F.pad loop, bmm / Linear /
LayerNorm) and, for the C++ stage, either
  * the reference's own extension compiled into oracle/_ref/extension.so ("reference"), or
  * the C restatement oracle/liboracle.so ("port") when that file is absent.
Unlike oracle.pyops (numpy, used for parity) this file is written for SPEED parity with the
reference: it is what "the reference cpp_extension CPU path timed on the host cores" means.
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import cabi, ref_import

_ext = None


def cpp_stage_kind():
    return "reference" if os.path.exists(ref_import._REF_SO) else "port"


def _ref_ext():
    global _ext
    if _ext is None:
        _ext = ref_import.load_ext()
    return _ext


def discretize(mid_feat, codebook):
    """discretization.py:58-70 with Adapter (seq-first [L+1, bs, D])."""
    seq = mid_feat[1:]
    n, bs, d = seq.shape
    idx = torch.cdist(seq.reshape(n * bs, d), codebook).argmin(dim=1)
    return idx.reshape(n, bs)


def pair_wise_point_sim(h, w, alpha=1.0, pow=2.0):
    i, j = torch.meshgrid(torch.arange(h, dtype=torch.float), torch.arange(w, dtype=torch.float), indexing="ij")
    p = torch.stack((i.flatten(), j.flatten()), dim=1)
    return 1 / (1 + torch.cdist(p, p, p=pow) / alpha)


def instance_graph(ingredients, attn, attn_cls, w_v, w_e, clamp=-1.0):
    """SchemaNet.forward (schema_net.py:377-399) -> three python lists."""
    attn_cls = attn_cls.masked_fill(attn_cls < clamp, float("-inf")).softmax(-1).nan_to_num(0)
    attn = torch.softmax(attn.masked_fill(attn < clamp, float("-inf")), dim=-1)
    geo = pair_wise_point_sim(14, 14)
    if cpp_stage_kind() == "reference":
        ext = _ref_ext()
        ids, v, num_v = ext.feat_to_instance_v(ingredients, attn_cls, w_v, True)
        sizes = num_v.tolist()
        inst_ids = list(torch.split_with_sizes(ids, sizes))
        inst_v = list(torch.split_with_sizes(v, sizes))
        dicts = [{v_: k for k, v_ in enumerate(i.tolist())} for i in inst_ids]
        inst_e = ext.feat_to_instance_e(ingredients, attn, geo, dicts, w_e, True, False)
        return inst_ids, inst_v, inst_e
    ids, _, v, num_v = cabi.instance_v(ingredients.numpy(), attn_cls.numpy(), w_v.numpy(), mean=True)
    splits = np.cumsum(num_v)[:-1]
    inst_ids = [torch.from_numpy(x) for x in np.split(ids, splits)]
    inst_v = [torch.from_numpy(x) for x in np.split(v, splits)]
    dicts = [{int(v_): k for k, v_ in enumerate(i.tolist())} for i in inst_ids]
    _, e = cabi.instance_e(ingredients.numpy(), attn.numpy(), geo.numpy(), dicts, w_e.numpy(), mean=True)
    return inst_ids, inst_v, [torch.from_numpy(x) for x in e]


def get_atlas(vertex_weights, edge_weights, thr=0.001):
    """schema_net.py:144-184 (edge_weights is pruned in place like the reference)."""
    def nsc(x, mn):
        x = x.clamp_min(mn)
        return (x / x.sum(-1, keepdim=True)).nan_to_num(0)
    cv = nsc(vertex_weights, 1.0e-5)
    mask = (cv > thr).float().unsqueeze(-1)
    mask = torch.bmm(mask, mask.transpose(1, 2))
    edge_weights.masked_fill_(~mask.bool(), 0)
    return cv, nsc(edge_weights * mask, 0)


def gnn(P, nodes, edges, ids, feat_mask=None):
    """gnn.py:78-98"""
    feat = F.embedding(ids, P["gnn.embedding.weight"])
    n_layers = len([k for k in P if k.endswith("g_conv.linear.weight")])
    for i in range(n_layers):
        adj = edges + edges.transpose(1, 2)
        In = torch.zeros_like(adj)
        In.diagonal(dim1=1, dim2=2).fill_(1)
        feat = torch.bmm(adj / 2 + In, feat)
        feat = F.linear(feat, P[f"gnn.layers.{i}.g_conv.linear.weight"], P[f"gnn.layers.{i}.g_conv.linear.bias"])
        if feat_mask is not None:
            feat.masked_fill_(feat_mask[..., None], 0)
        feat = F.relu(F.layer_norm(feat, feat.shape[-1:], P[f"gnn.layers.{i}.norm.weight"], P[f"gnn.layers.{i}.norm.bias"]))
    feat = (feat * nodes[..., None]).mean(dim=1)
    return F.linear(feat, P["gnn.fc.weight"], P["gnn.fc.bias"])


def matcher(P, inst_ids, inst_v, inst_e, cv, ce, class_ingredients, num_codes):
    """match.py:33-76, inner_product"""
    bs = len(inst_ids)
    sizes = [len(x) for x in inst_ids]
    n = max(sizes)
    mask = torch.zeros(bs, n, dtype=torch.bool)
    for i, s in enumerate(sizes):
        mask[i, s:].fill_(1)
        inst_ids[i] = F.pad(inst_ids[i], (0, n - s), value=num_codes)
        inst_v[i] = F.pad(inst_v[i], (0, n - s))
        inst_e[i] = F.pad(inst_e[i], (0, n - s, 0, n - s))
    fi = gnn(P, torch.stack(inst_v), torch.stack(inst_e), torch.stack(inst_ids), mask)
    fk = gnn(P, cv, ce, class_ingredients)
    return (fi.unsqueeze(1) * fk.unsqueeze(0)).sum(-1)


@torch.no_grad()
def forward(tokens_bf, attn_full, codebook, vertex_weights, edge_weights, class_ingredients, P, w_v, w_e, also_fp64=False):
    """tokens_bf [B, L+1, D] batch-first, attn_full [B, L+1, L+1] head-averaged logits.
    Returns (pred [B, K], word ids, per-stage seconds).  also_fp64: the matcher (GNN on instances and on the atlas,
    similarity) is run a second time in float64 on the same fp32 graphs - the value the reference's fp32 arithmetic and
    the HIP path both approximate; stages["pred_fp64"] holds it (parity tests: |hip - fp64| <= |fp32 - fp64| + slack)."""
    t = [time.perf_counter()]
    mid = tokens_bf.transpose(0, 1).contiguous()                       # backbone layout [L+1, bs, D]
    ing = discretize(mid, codebook).transpose(0, 1).contiguous()
    t.append(time.perf_counter())
    attn = attn_full[:, 1:, 1:].contiguous()
    attn_cls = attn_full[:, 0, 1:].contiguous()
    inst_ids, inst_v, inst_e = instance_graph(ing, attn, attn_cls, w_v, w_e)
    t.append(time.perf_counter())
    cv, ce = get_atlas(vertex_weights, edge_weights.clone())
    t.append(time.perf_counter())
    pred64 = None
    if also_fp64:                                                      # (before the fp32 call: `matcher` pads its lists in place)
        P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
        pred64 = matcher(P64, [x.clone() for x in inst_ids], [x.double() for x in inst_v], [x.double() for x in inst_e],
                         cv.double(), ce.double(), class_ingredients, codebook.shape[0])
        t[-1] = time.perf_counter()
    pred = matcher(P, inst_ids, inst_v, inst_e, cv, ce, class_ingredients, codebook.shape[0])
    t.append(time.perf_counter())
    stages = dict(zip(("discretize", "instance_graph", "atlas", "match"), np.diff(t).tolist()))
    if also_fp64:
        stages["pred_fp64"] = pred64
    return pred, ing, stages
