"""CPU: the oracle (oracle/schemanet_oracle.c + oracle/pyops.py) against the fixtures the real
reference produced (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import datagen
from oracle import cabi, pyops

RTOL = 1e-5   # north_star tolerance for floating-point scores


def close(a, b, rtol=RTOL, atol=1e-7):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ G1
def test_assign_matches_reference_indices(golden):
    g = golden("assign.npz")
    seq, ing = pyops.discretize(g["mid_feat"], g["codebook"], activate=True)
    assert ing.dtype == np.int64 and ing.shape == g["ingredients"].shape
    assert np.array_equal(ing, g["ingredients"])          # bit-exact, incl. tie + near-tie rows
    assert ing[5, 0] == 3                                   # duplicate codewords 3 / 7: first wins
    assert {int(ing[9, 0]), int(ing[9, 1])} == {20, 21}     # near-tie resolved on both sides
    assert np.array_equal(seq, g["seq_active"])
    seq2, _ = pyops.discretize(g["mid_feat"], g["codebook"], activate=False)
    assert bool(g["seq_inactive_equals_input"]) and np.array_equal(seq2, g["mid_feat"])


# ------------------------------------------------------------------ ext level (bit-exact)
def test_ext_small_instance_v_bit_exact(golden):
    g = golden("ext_small.npz")
    ids, a2, w, num_v = cabi.instance_v(g["ing"], g["attn_cls"], g["w_v"], mean=True)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(num_v, g["num_v"])
    close(w, g["weights"], rtol=2e-7)
    _, _, ws, _ = cabi.instance_v(g["ing"], g["attn_cls"], g["w_v"], mean=False)
    close(ws, g["weights_sum"], rtol=2e-7)
    # ids are exactly np.unique per image
    o = 0
    for b, n in enumerate(num_v.tolist()):
        assert np.array_equal(ids[o:o + n], np.unique(g["ing"][b]))
        o += n


def _dicts(ids, num_v):
    out, o = [], 0
    for n in num_v.tolist():
        out.append({int(v): k for k, v in enumerate(ids[o:o + n].tolist())})
        o += n
    return out


def test_ext_small_instance_e(golden):
    g = golden("ext_small.npz")
    dicts = _dicts(g["ids"], g["num_v"])
    _, e = cabi.instance_e(g["ing"], g["attn"], g["geo"], dicts, g["w_e"], mean=True)
    flat = np.concatenate([x.reshape(-1) for x in e])
    close(flat, g["edges"], rtol=1e-6)
    _, es = cabi.instance_e(g["ing"], g["attn"], g["geo"], dicts, g["w_e"], mean=False)
    close(np.concatenate([x.reshape(-1) for x in es]), g["edges_sum"], rtol=1e-6)
    # NaN semantics (SURVEY hard part 3): image 3 has a NaN attention source row; only the attn
    # channel of that word's row is zeroed, the geo channel survives
    e2, _ = cabi.instance_e(g["ing"], g["attn"], g["geo"], dicts, g["w_e"], mean=True)
    row = dicts[3][int(g["ing"][3, 2])]
    assert np.all(e2[3][row, :, 1] == 0) and np.all(e2[3][row, :, 0] > 0)
    assert bool(g["rsl_raises"])      # reference raises for remove_self_loop=True (documented)
    # non-canonical dictionary (permuted rows, extra key, missing word -> slot 0)
    odd = {int(k): int(v) for k, v in zip(g["odd_keys"], g["odd_vals"])}
    _, eo = cabi.instance_e(g["ing"][:1], g["attn"][:1], g["geo"], [odd], g["w_e"], mean=True)
    close(eo[0], g["edges_odd0"], rtol=1e-6)


def test_ext_small_dense_init_functions_bit_exact(golden):
    g = golden("ext_small.npz")
    M = g["v_attr"].shape[1]
    assert np.array_equal(cabi.v_attr(g["ing"], g["attn_cls"], M, mean=True), g["v_attr"])
    assert np.array_equal(cabi.v_attr(g["ing"], g["attn_cls"], M, mean=False), g["v_attr_sum"])
    assert np.array_equal(cabi.v_attr(g["ing"], g["attn_cls"], M, True, True), g["v_attr_io"])
    cdict = [{int(k): v for v, k in enumerate(row)} for row in g["class_ingredients"]]
    tab = cabi.dicts_to_slot_table(cdict, M)
    fe = cabi.feat_to_e(g["ing"], g["attn"], g["geo"], tab, g["label"], g["feat_to_e"].shape[1])
    assert np.array_equal(np.isnan(fe), np.isnan(g["feat_to_e"]))
    assert np.array_equal(np.nan_to_num(fe), np.nan_to_num(g["feat_to_e"]))   # same sum order


def test_geo_table_bit_exact(golden):
    g = golden("ext_small.npz")
    assert np.array_equal(pyops.pair_wise_point_sim(6, 6, 1.0, 2.0), g["geo"])


# ------------------------------------------------------------------ G2 G3
def test_instance_graph_from_logits(golden):
    g = golden("graph.npz")
    B, L, M, seed = g["case"].tolist()
    ing, attn, attn_cls = datagen.graph_case(B, L, M, seed)
    out = pyops.instance_graph(ing, attn, attn_cls, g["w_v"], g["w_e"])
    assert np.array_equal(out["num_v"], g["num_v"])
    assert g["num_v"].tolist()[1:3] == [1, L]               # all-same and all-distinct images
    assert np.array_equal(np.concatenate(out["instance_ingredients"]), g["ids"])
    close(np.concatenate(out["instance_vertices"]), g["vertices"])
    close(np.concatenate([e.reshape(-1) for e in out["instance_edges"]]), g["edges"])
    # in-place side effect on the caller's attn_cls (schema_net.py:296)
    assert np.array_equal(out["attn_cls_masked"], g["attn_cls_after"])
    assert bool(g["attn_after_row17_isinf"]) and np.isinf(out["attn_masked"][0, 17]).all()


# ------------------------------------------------------------------ G4 G7
def test_init_path_statistics(golden):
    g = golden("graph.npz")
    B, L, M, seed, K, n_max = g["init_case"].tolist()
    ing, attn, attn_cls = datagen.graph_case(B, L, M, seed)
    w = np.full((2, 1), 0.5, np.float32)                    # SchemaNet default (schema_net.py:106)
    fv = pyops.full_vertices(ing, attn_cls, M, w)
    assert np.array_equal(np.isnan(fv), np.isnan(g["full_vertices"]))
    close(np.nan_to_num(fv), np.nan_to_num(g["full_vertices"]))
    cv, sums, n = pyops.init_class_vertices(ing, attn_cls, g["init_label"], K, M, w)
    assert np.array_equal(n, g["n_tracked"])
    close(np.nan_to_num(sums), np.nan_to_num(g["class_vertex_sums"]))
    close(np.nan_to_num(cv), np.nan_to_num(g["class_vertices_norm"]), rtol=2e-5)
    cdict = [{int(k): v for v, k in enumerate(row)} for row in g["topk_index"]]
    tab = cabi.dicts_to_slot_table(cdict, M)
    fe = pyops.limited_edges(ing, attn, g["init_label"], tab, n_max, w)
    close(fe, g["limited_edges"])
    ew, esum, _ = pyops.init_graph(ing, attn, g["init_label"], tab, K, n_max, w)
    close(esum, g["class_edge_sums"])
    assert np.array_equal(np.isnan(ew), np.isnan(g["class_edge_mean"]))   # never-seen class: 0/0


# ------------------------------------------------------------------ G5 G6
def test_atlas_and_matcher(golden):
    g = golden("matcher.npz")
    B, L, M, seed, K, n_max, E = g["case"].tolist()
    cv, ce, ew_after = pyops.get_atlas(g["vertex_weights"], g["edge_weights"], 0.001)
    close(cv, g["class_vertices"], rtol=2e-6)
    close(ce, g["class_edges"], rtol=2e-6)
    assert np.array_equal(ew_after, g["edge_weights_after"])          # in-place prune side effect
    assert (g["class_vertices"] <= 0.001).any() and (ce[2, 5] == 0).all()
    params = {k[6:]: v for k, v in g.items() if k.startswith("param:")}
    ing, attn, attn_cls = datagen.graph_case(B, L, M, seed)
    w = np.full((2, 1), 0.5, np.float32)
    inst = pyops.instance_graph(ing, attn, attn_cls, w, w)
    for sim in ("inner_product", "cosine", "euclidean"):
        pred = pyops.matcher_forward(params, inst["instance_ingredients"], inst["instance_vertices"],
                                     inst["instance_edges"], cv, ce, g["class_ingredients"], M, sim)
        close(pred, g["pred_" + sim], rtol=2e-5, atol=1e-6)
    # batch-dependent pooling (gnn.py:96): solo batches differ from the joint batch
    solo = np.stack([
        pyops.matcher_forward(params, inst["instance_ingredients"][b:b + 1], inst["instance_vertices"][b:b + 1],
                              inst["instance_edges"][b:b + 1], cv, ce, g["class_ingredients"], M)[0]
        for b in range(B)])
    close(solo, g["pred_inner_product_solo"], rtol=2e-5, atol=1e-6)
    assert not np.allclose(solo[1], g["pred_inner_product"][1], rtol=1e-3)


# ------------------------------------------------------------------ G8
def test_wrapper_head_mean(golden):
    g = golden("wrapper.npz")
    bs, H, L, seed = g["case"].tolist()
    extracted = datagen.bellish((bs * H, L + 1, L + 1), seed, 2.0)
    attn, attn_cls = pyops.wrapper_attention(extracted, bs)
    close(attn, g["attn"], rtol=1e-6)
    close(attn_cls, g["attn_cls"], rtol=1e-6)


# ------------------------------------------------------------------ properties
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_properties_random(seed):
    B, L, M = 3, 49, 40
    ing = datagen.integers((B, L), seed, M)
    sm = pyops.clamp_softmax(datagen.bellish((B, L, L), seed + 50, 1.5), -1.0, False)[1]
    smc = pyops.clamp_softmax(datagen.bellish((B, L), seed + 60, 1.5), -1.0, True)[1]
    w = np.asarray([[0.25], [0.75]], np.float32)
    ids, a2, wt, num_v = cabi.instance_v(ing, smc, w)
    # permutation of positions leaves vertex attrs unchanged up to summation order
    perm = np.argsort(datagen.hash_u24((L,), seed + 70))
    ids_p, a2_p, _, num_p = cabi.instance_v(ing[:, perm], smc[:, perm], w)
    assert np.array_equal(ids, ids_p) and np.array_equal(num_v, num_p)
    close(a2, a2_p, rtol=1e-6)
    assert (a2.max() <= 1.0 + 1e-6)
    geo = pyops.pair_wise_point_sim(7, 7)
    e2, e = cabi.instance_e(ing, sm, geo, _dicts(ids, num_v), w)
    for x2, x in zip(e2, e):
        close(x2.sum(axis=1), np.ones_like(x2.sum(axis=1)), rtol=1e-5)    # each channel row-normalised
        close(x.sum(axis=1), np.full(x.shape[0], w.sum()), rtol=1e-5)     # rows sum to w0+w1


# ------------------------------------------------------------------------------- codebook extraction
def _kmeans_case(seed, k, D, N, dup=False):
    rng = np.random.default_rng(seed)
    centres = (rng.normal(size=(k, D)) * 2).astype(np.float32)
    x = (centres[rng.integers(0, k, N)] + rng.normal(size=(N, D)).astype(np.float32)).astype(np.float32)
    guess = x[rng.choice(N, k, replace=False)].copy()
    if dup:
        guess[3] = guess[1]            # a duplicated centre never wins a tie (first index) -> no members -> dropped
    return x, guess


@pytest.mark.parametrize("seed,k,D,N,dup", [(0, 16, 64, 3000, False), (1, 24, 192, 4000, True)])
def test_kmeans_oracle_equals_scipy(seed, k, D, N, dup):
    """The k-means restatement (exact nearest centre + SciPy's fp32 in-order member sums + its stopping
    rule) reproduces scipy.cluster.vq.kmeans - the call of the reference's extract_ingredients.py:33-36 -
    bit for bit, including the dropped empty cluster."""
    from scipy.cluster.vq import kmeans
    x, guess = _kmeans_case(seed, k, D, N, dup)
    want, want_dist = kmeans(x, guess, thresh=1e-5)
    got, got_dist, iters = pyops.kmeans_lloyd(x, guess, thresh=1e-5)
    assert got.dtype == np.float32 and got.shape == want.shape and (not dup or got.shape[0] == k - 1)
    assert np.array_equal(got, want)
    assert abs(got_dist - float(want_dist)) <= 1e-5 * float(want_dist) and iters >= 2


def test_loss_oracle_matches_reference_fixture(golden):
    """The loss restatement against the loss terms the real reference produced for the train-step fixture
    (atlas = normalize() then get_atlas() of the fixture's SchemaNet state, both restated in numpy)."""
    g = golden("train_step.npz")
    vw = np.clip(g["sn:vertex_weights.tensor"], 0.0, None)
    vw = np.nan_to_num(vw / vw.sum(-1, keepdims=True))                      # SchemaNet.normalize: normalize_sum_ after clamp_min(0)
    ew = np.clip(g["sn:edge_weights.tensor"], 0.0, None)
    ew = np.nan_to_num(ew / ew.sum(-1, keepdims=True))
    cv, ce, _ = pyops.get_atlas(vw.astype(np.float32), ew.astype(np.float32), 0.001)
    got = pyops.schema_inference_loss(g["pred"], g["label"], cv, ce)
    for k, v in got.items():
        np.testing.assert_allclose(v, g["loss:" + k], rtol=2e-5, err_msg=k)
