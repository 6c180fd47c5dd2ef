"""GPU (-m gpu): the configurations of BASELINE.json at their REAL sizes.

  * config [3] DeiT-Base + ImageNet-1k shape on one GPU: D = 768 bf16 tokens, 1024 words, K = 1000 classes of 500
    vertices (a 1 GB IR-Atlas), GNN width 1024, 256 images - word ids against the oracle on a sample of 4 096 tokens
    and against the exact kernel on all of them, scores against oracle/cpu_pipeline.py (the reference's forward op for
    op on the host) on a sample of 8 images x all 1000 classes, size-independent properties on the rest;
  * an S1 census: 10 million tokens of the config [1] shape against the reference's own arithmetic,
    torch.cdist(...).argmin (reference discretization/discretization.py:65) on the host.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cabi
from test_gpu_parity import DEV, RTOL, as_good_as_fp32_reference, mods, scores_close  # noqa: F401  (mods: fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, payload):
    """best effort: numbers a reader may want next to the pass/fail (gpurun merges gpurun_out/ back)"""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(payload, fh, indent=1)
    except OSError:
        pass


# =============================================================================== config [3] at its real size
def test_c4_real_size_one_gpu(mods):
    from oracle import cpu_pipeline
    graph, disc_mod, ops = mods["graph"], mods["disc"], mods["ops"]
    B, D, M, K, E, L, n_max = 256, 768, 1024, 1000, 1024, 196, 500
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    pool = torch.randn(4 * M, D, generator=g(1))
    codebook = pool[torch.randperm(4 * M, generator=g(11))[:M]] + 0.05 * torch.randn(M, D, generator=g(2))
    tokens = torch.randn(B, L + 1, D, generator=g(3))
    near = torch.randint(0, M, (B, L), generator=g(4))
    tokens[::2, 1:] = codebook[near[::2]] + 0.3 * tokens[::2, 1:]            # half the images k-means like
    tokens = tokens.to(torch.bfloat16)                                       # what an AMP backbone hands over
    attn = torch.randn(B, L + 1, L + 1, generator=g(5))
    torch.manual_seed(6)
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, dist_pow=2, feat_h=14, feat_w=14, clamp_vertex_attn=-1.0,
                         clamp_edge_attn=-1.0, remove_self_loop=False, prune_node_threshold=0.001, class_max_vertices=n_max)
    perm = torch.stack([torch.randperm(M, generator=g(20 + k))[:n_max] for k in range(K)])
    sn.register_class_vertices(perm)
    with torch.no_grad():
        sn.vertex_weights.tensor[:, ::9] = 0.0                               # pruned vertices
        sn.vertex_attribute_weights.tensor.copy_(torch.tensor([[0.3], [0.7]]))
        sn.edge_attribute_weights.tensor.copy_(torch.tensor([[0.6], [0.4]]))
    torch.manual_seed(7)
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    with torch.no_grad():
        for layer in m.gnn.layers:
            layer.norm.weight.uniform_(0.5, 1.5); layer.norm.bias.uniform_(-0.5, 0.5)
        m.gnn.fc.bias.zero_()                                                # (its initial value; the scaling property below needs it)
    disc = disc_mod.Discretization(M, D)
    with torch.no_grad():
        disc.vocabulary.weight.copy_(codebook)
    sample = torch.tensor([0, 1, 37, 100, 101, 202, 254, 255])
    P = {"gnn." + k: v.detach().clone() for k, v in m.gnn.state_dict().items()}
    want, ing_cpu, stages = cpu_pipeline.forward(tokens[sample].float(), attn[sample], codebook, sn.vertex_weights.tensor.detach().clone(),
                                                 sn.edge_weights.tensor.detach().clone(), sn.class_ingredients.tensor.clone(), P,
                                                 sn.vertex_attribute_weights.tensor.detach().clone(), sn.edge_attribute_weights.tensor.detach().clone(),
                                                 also_fp64=True)
    disc, sn, m = disc.to(DEV), sn.to(DEV), m.to(DEV)
    tok_d, attn_d = tokens.to(DEV), attn.to(DEV)

    def run(tk, at):
        with torch.no_grad():
            atlas = m.atlas_features_async(lambda: sn.get_atlas(fused_adjacency=True))     # E = 1024: the wide MFMA route
            ing = disc.assign(tk[:, 1:, :])
            gr = sn.instance_graph_padded(ing, at[:, 1:, 1:], at[:, 0, 1:], mutate_inputs=False)
            return ing, gr, m.forward_padded(gr, atlas.class_dict, feat_kg=atlas)

    ing_full, gr_full, pred_full = run(tok_d, attn_d)
    ing_sub, gr_sub, pred_sub = run(tok_d[sample.to(DEV)].contiguous(), attn_d[sample.to(DEV)].contiguous())
    torch.cuda.synchronize()
    # ---- word ids: the oracle on a sample of 4 096 tokens, the exact kernel on all 50 176, torch.cdist on the 8 images
    flat = tokens[:, 1:].float().reshape(-1, D).numpy()
    pick = np.random.default_rng(0).choice(flat.shape[0], 4096, replace=False)
    want_ids = cabi.assign_words(np.ascontiguousarray(flat[pick]), codebook.numpy())
    assert np.array_equal(ing_full.cpu().numpy().reshape(-1)[pick], want_ids)
    cbt, packed = ops.PackedCodebook().get(disc.vocabulary.weight)
    exact = ops.assign_words(tok_d[:, 1:, :], cbt, packed, mode=1)
    assert torch.equal(ing_full, exact)
    assert torch.equal(ing_sub.cpu(), ing_cpu) and torch.equal(ing_full[sample.to(DEV)], ing_sub)
    # ---- scores of the 8 sampled images x 1000 classes against the reference forward on the host
    assert tuple(pred_full.shape) == (B, K) and tuple(pred_sub.shape) == (len(sample), K)
    rel = scores_close(pred_sub, want, "C4 real size, 8 images x 1000 classes")
    vs64 = as_good_as_fp32_reference(pred_sub, want, stages["pred_fp64"], "C4 real size, 8 images x 1000 classes")
    scale = want.abs().max().item()
    top2 = want.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 4 * RTOL * scale
    assert torch.equal(pred_sub.cpu().argmax(1)[decided], want.argmax(1)[decided])
    # ---- properties on the full batch: finite; graphs of the sampled images identical in either batch; and since the
    # pooling divides by the padded length of the batch (gnn.py:96) and fc.bias = 0, the scores of an image scale with
    # n_max(sub-batch) / n_max(full batch) and its top-1 does not depend on its batch mates
    assert torch.isfinite(pred_full).all()
    n_full, n_sub = int(gr_full["n_max"].item()), int(gr_sub["n_max"].item())
    assert n_sub <= n_full and torch.equal(gr_full["n"][sample.to(DEV)], gr_sub["n"])
    rescaled = pred_full[sample.to(DEV)].cpu().double() * n_full / n_sub
    err = (rescaled - pred_sub.cpu().double()).abs().max().item()
    assert err <= RTOL * scale, (err, scale)
    assert torch.equal(pred_full[sample.to(DEV)].argmax(1)[decided.to(DEV)], pred_sub.argmax(1)[decided.to(DEV)])
    _report("c4_real_size.json", {"images": B, "classes": K, "vertices_per_class": n_max, "gnn_width": E, "score_scale": scale, "vs_fp64": vs64,
                                  "max_elementwise_rel_err_above_floor": rel, "n_max_full": n_full, "n_max_sample": n_sub})


# =============================================================================== S1 census against torch.cdist
def test_s1_census_against_torch_cdist(mods):
    """>= 10 M tokens of the config [1] shape (D = 384, 512 words; batches alternate between k-means-like tokens and
    plain normal ones): HIP ids vs `torch.cdist(x, codebook).argmin(1)` in fp32 on the host - the reference's own
    arithmetic (discretization.py:65).  The HIP path returns the TRUE nearest word (fp64 re-rank, lowest index on exact
    ties); cdist's fp32 `|x|^2 + |c|^2 - 2 x.c` can pick another word only inside its own rounding error.  Every mismatch
    must therefore be (a) a word that is not nearer in fp64 than the HIP one and (b) within the fp32 error bound of the
    expansion: gamma (|x| + |c|)^2 with gamma = 2 (D + 2) 2^-24, plus the sqrt rounding 4 x 2^-24 d^2."""
    disc_mod = mods["disc"]
    B, L, D, M = 256, 196, 384, 512
    n_batches = int(os.environ.get("SN_CENSUS_BATCHES", "200"))              # 200 x 50 176 = 10 035 200 tokens
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    pool = torch.randn(4096, D, generator=g(1))
    codebook = pool[torch.randperm(4096, generator=g(11))[:M]] + 0.05 * torch.randn(M, D, generator=g(2))
    disc = disc_mod.Discretization(M, D).to(DEV)
    with torch.no_grad():
        disc.vocabulary.weight.copy_(codebook)
    cb_d, cb64 = codebook.to(DEV), codebook.double()
    gen = torch.Generator(device=DEV)
    n_tok, mism, worst_margin, worst_ratio, hip_worse = 0, 0, 0.0, 0.0, 0
    per_kind = {"kmeans_like": [0, 0], "normal": [0, 0]}
    for i in range(n_batches):
        gen.manual_seed(1000 + i)
        x = torch.randn(B, L, D, generator=gen, device=DEV)
        kind = "kmeans_like" if i % 2 == 0 else "normal"
        if kind == "kmeans_like":
            near = torch.randint(0, M, (B, L), generator=gen, device=DEV)
            x = cb_d[near] + (0.3 + 0.1 * (i % 7)) * x
        with torch.no_grad():
            ids = disc.assign(x).reshape(-1).cpu()
        xc = x.reshape(-1, D).cpu()
        ref = torch.cdist(xc, codebook).argmin(dim=1)
        bad = (ids != ref).nonzero().reshape(-1)
        n_tok += xc.shape[0]
        per_kind[kind][0] += xc.shape[0]
        per_kind[kind][1] += int(bad.numel())
        mism += int(bad.numel())
        if bad.numel():
            xb = xc[bad].double()
            d_h = ((xb - cb64[ids[bad]]) ** 2).sum(1)
            d_r = ((xb - cb64[ref[bad]]) ** 2).sum(1)
            margin = d_r - d_h                                               # >= 0: the HIP word is at least as near
            xn, cn_h, cn_r = xb.norm(dim=1), cb64[ids[bad]].norm(dim=1), cb64[ref[bad]].norm(dim=1)
            bound = 2.0 * (D + 2) * 2.0 ** -24 * (xn + torch.maximum(cn_h, cn_r)) ** 2 + 4 * 2.0 ** -24 * d_r
            hip_worse += int((margin < 0).sum())
            assert (margin >= 0).all(), f"batch {i}: the HIP word is farther than cdist's in fp64 (margin {margin.min().item():.3e})"
            assert (margin <= bound).all(), f"batch {i}: a mismatch of {margin.max().item():.3e} exceeds cdist's fp32 error bound {bound.min().item():.3e}"
            worst_margin = max(worst_margin, float(margin.max()))
            worst_ratio = max(worst_ratio, float((margin / bound).max()))
    assert n_tok >= 10_000_000 or n_batches < 200
    _report("s1_census.json", {"tokens": n_tok, "mismatches_vs_torch_cdist_argmin": mism, "mismatch_rate": mism / n_tok,
                               "hip_word_farther_in_fp64": hip_worse, "largest_fp64_margin_of_a_mismatch_sq_dist": worst_margin,
                               "largest_margin_over_cdist_fp32_bound": worst_ratio,
                               "per_kind_tokens_mismatches": per_kind, "shape": {"D": D, "M": M, "tokens_per_batch": B * L}})
