"""GPU parity of the deferred S1 finish (round 4): `sn_assign_words(mode 2)` runs the fp16 screen only and the
instance-graph kernel finishes the undecided tokens of its image in its row phase (csrc/sn_graph.hip, RerankWave /
rerank_overflow_token).  The ids must be those of mode 0 and of the oracle bit for bit, and the graph built on them the
one built on final ids.  Reference op: discretization/discretization.py:65 (torch.cdist + argmin); consumer
schema_net.py:278-356."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "schemanet-pytorch_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import datagen  # noqa: E402
from oracle import cabi  # noqa: E402
from test_gpu_parity import DEV, T, make_schema_net, mods  # noqa: E402,F401  (mods: fixture)

pytestmark = pytest.mark.gpu


def _flags(handle, n_tok):
    return handle.ws[32:32 + 4 * n_tok].view(torch.int32)


def _case(D, M, B, L, seed, hard=True):
    """codebook, batch-first tokens [B, L + 1, D] (row 0 = cls) with the cases the finish must get right: isotropic tokens
    (several candidates inside the fp16 window), k-means-like ones (nothing flagged), exact ties, tokens in the middle of
    four words of one accumulator group (overflow: a fourth candidate hides behind a full triple), and a non-finite token
    (the screen cannot bound it: every word is a candidate)."""
    cb = datagen.bellish((M, D), seed, 1.0)
    tok = datagen.bellish((B, L + 1, D), seed + 1, 1.0)
    near = datagen.integers((B, L), seed + 2, M)
    tok[1, 1:] = cb[near[1]] + 0.3 * tok[1, 1:]                       # k-means like image
    if hard:
        tok[2, 5] = 0.5 * (cb[3] + cb[77 % M])                        # exact tie of two words: lowest index wins
        for j, m0 in enumerate((1, 6, 11)):                           # words m0, m0 + 32, m0 + 64, m0 + 96: same row of four tiles = one group
            if m0 + 96 < M:
                tok[3, 10 + 7 * j] = 0.25 * (cb[m0] + cb[m0 + 32] + cb[m0 + 64] + cb[m0 + 96]) + 1e-4 * tok[3, 10 + 7 * j]
                tok[0, 20 + j] = 0.25 * (cb[m0] + cb[m0 + 32] + cb[m0 + 64] + cb[m0 + 96])          # the exact centroid: four-way tie
        tok[4 % B, 9, 0] = np.inf
        tok[4 % B, 11, 3] = np.nan
        tok[5 % B, 50] = 4.0e4                                        # beyond what goes through fp16
    return cb, tok


@pytest.fixture(params=[0, 5], ids=["token-stationary", "k-outer"])
def screen_form(request, mods):
    """both forms of the screen write the records the finish reads (the K-outer form applies to 16-tile codebooks; elsewhere it
    falls back to the default form)"""
    lib = mods["cx"].load()
    old = lib.sn_assign_variant()
    assert lib.sn_assign_set_variant(request.param) == 0
    yield request.param
    lib.sn_assign_set_variant(old)


@pytest.mark.parametrize("D,M", [(384, 512), (192, 128), (384, 1000), (192, 480), (768, 1000)])
@pytest.mark.parametrize("layout", ["batch_first", "sequence_first"])
def test_deferred_finish_equals_mode0_and_oracle(mods, D, M, layout, screen_form):
    ops, lib = mods["ops"], mods["cx"].load()
    B, L, K = 8, 196, 5
    cb, tok = _case(D, M, B, L, seed=900 + D + M)
    attn = datagen.bellish((B, L + 1, L + 1), 77, 1.5)
    cbt, packed = ops.PackedCodebook().get(T(cb))
    if layout == "batch_first":
        tok_d = T(tok)
        x = tok_d[:, 1:, :]                                           # [B, L, D]
        view = lambda ids: ids                                        # noqa: E731
    else:
        tok_d = T(np.ascontiguousarray(tok.transpose(1, 0, 2)))       # [L + 1, B, D]
        x = tok_d[1:]                                                 # [L, B, D]
        view = lambda ids: ids.t()                                    # noqa: E731
    assert lib.sn_assign_defers(M, D) == 1
    want = ops.assign_words(x, cbt, packed)                           # mode 0: screen + stand-alone re-rank
    oracle = cabi.assign_words(x.cpu().numpy().reshape(-1, D), cb).reshape(tuple(x.shape[:2]))
    assert np.array_equal(want.cpu().numpy(), oracle)
    # ---- the deferred form: screen only, then (a) the stand-alone finish, (b) the finish inside the graph kernel
    ids_a, h_a = ops.assign_words(x, cbt, packed, defer=True)
    assert h_a is not None
    fl = _flags(h_a, x.shape[0] * x.shape[1])
    n_flagged, n_over, n_full = int((fl > 0).sum()), int((fl < 0).sum()), int(((fl < 0) & ((fl & 0x40000000) != 0)).sum())
    assert n_flagged > 0 and n_over > 0 and n_full >= 2, (n_flagged, n_over, n_full)      # the case holds what it is built for
    tentative = ids_a.clone()
    h_a.finish()
    assert torch.equal(ids_a, want) and not torch.equal(tentative, want)
    ids_b, h_b = ops.assign_words(x, cbt, packed, defer=True)
    sn = make_schema_net(mods, M, K)
    a_d = T(attn)
    g_ref = sn.instance_graph_padded(view(want).contiguous(), a_d[:, 1:, 1:], a_d[:, 0, 1:], mutate_inputs=False, zero_padding=False)
    g_fused = sn.instance_graph_padded(view(ids_b), a_d[:, 1:, 1:], a_d[:, 0, 1:], mutate_inputs=False, zero_padding=False, rerank=h_b)
    torch.cuda.synchronize()
    assert h_b.done
    assert torch.equal(ids_b, want), int((ids_b != want).sum())      # written back: final ids
    assert torch.equal(g_fused["n"], g_ref["n"]) and torch.equal(g_fused["n_max"], g_ref["n_max"])
    assert torch.equal(g_fused["ids"], g_ref["ids"])
    assert torch.equal(g_fused["vertices"], g_ref["vertices"])
    n = g_ref["n"].tolist()
    for b in range(B):                                                # (rows / columns beyond the vertex count are unwritten)
        assert torch.equal(g_fused["edges"][b, :n[b], :n[b]], g_ref["edges"][b, :n[b], :n[b]]), b


def test_deferred_finish_with_per_head_taps_and_fallbacks(mods):
    """the API's input (raw per-head attention taps: the static-row form of the kernel) takes the finish as well; a
    consumer that is not the prediction configuration (zero padding asked for) gets the stand-alone finish instead; shapes
    without a deferred form hand back no handle"""
    ops, lib = mods["ops"], mods["cx"].load()
    D, M, B, L, H, K = 384, 512, 6, 196, 3, 5
    cb, tok = _case(D, M, B, L, seed=1234)
    cbt, packed = ops.PackedCodebook().get(T(cb))
    x = T(tok)[:, 1:, :]
    want = ops.assign_words(x, cbt, packed)
    ext = T(datagen.bellish((B, H, L + 1, L + 1), 78, 2.0))
    sn = make_schema_net(mods, M, K)
    args = (ext[:, :, 1:, 1:], ext[:, :, 0, 1:])
    g_ref = sn.instance_graph_padded(want, *args, mutate_inputs=False, zero_padding=False)
    ids, h = ops.assign_words(x, cbt, packed, defer=True)
    g = sn.instance_graph_padded(ids, *args, mutate_inputs=False, zero_padding=False, rerank=h)
    assert h.done and torch.equal(ids, want) and torch.equal(g["ids"], g_ref["ids"]) and torch.equal(g["vertices"], g_ref["vertices"])
    # zero padding: the general kernel - the handle is finished by the stand-alone kernels first
    ids2, h2 = ops.assign_words(x, cbt, packed, defer=True)
    g2 = sn.instance_graph_padded(ids2, *args, mutate_inputs=False, zero_padding=True, rerank=h2)
    assert h2.done and torch.equal(ids2, want) and torch.equal(g2["ids"], g_ref["ids"])
    # no deferred form: codebooks of more than 2048 words (16-bit word codes), widths other than 192 / 384 / 768
    assert lib.sn_assign_defers(1024, 768) == 1 and lib.sn_assign_defers(4096, 384) == 0 and lib.sn_assign_defers(512, 256) == 0
    cb3 = T(datagen.bellish((2304, 384), 5, 1.0))
    cbt3, packed3 = ops.PackedCodebook().get(cb3)
    x3 = T(datagen.bellish((2, L, 384), 6, 1.0))
    ids3, h3 = ops.assign_words(x3, cbt3, packed3, defer=True)
    assert h3 is None and torch.equal(ids3, ops.assign_words(x3, cbt3, packed3))
    # the K-outer screen writes the default screen's records: the same deferred finish
    old = lib.sn_assign_variant()
    try:
        lib.sn_assign_set_variant(5)
        assert lib.sn_assign_defers(512, 384) == 1
        ids4, h4 = ops.assign_words(x, cbt, packed, defer=True)
        g4 = sn.instance_graph_padded(ids4, *args, mutate_inputs=False, zero_padding=False, rerank=h4)
        assert h4 is not None and h4.done and torch.equal(ids4, want) and torch.equal(g4["ids"], g_ref["ids"])
    finally:
        lib.sn_assign_set_variant(old)


def test_predictor_and_bench_step_take_the_deferred_finish(mods):
    """`SchemaNetPredictor.forward` (eager and replayed) and bench.py's step give the scores of the non-deferred route bit
    for bit (SN_S1_DEFER=0 = the stand-alone re-rank of rounds 1-3)."""
    import bench
    graph, disc_mod = mods["graph"], mods["disc"]
    B, L, D, M, K, E = 6, 196, 384, 512, 7, 256
    cb, tok = _case(D, M, B, L, seed=4321, hard=False)
    attn = datagen.bellish((B, L + 1, L + 1), 79, 1.5)
    disc = disc_mod.Discretization(M, D).to(DEV)
    with torch.no_grad():
        disc.vocabulary.weight.copy_(T(cb))
    torch.manual_seed(1)
    sn = make_schema_net(mods, M, K)
    sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)
    tok_d, attn_d = T(tok), T(attn)
    old = os.environ.get("SN_S1_DEFER")
    try:
        with torch.no_grad():
            os.environ["SN_S1_DEFER"] = "0"
            ref = bench.step(disc, sn, m, tok_d, attn_d).clone()
            os.environ["SN_S1_DEFER"] = "1"
            got = bench.step(disc, sn, m, tok_d, attn_d).clone()
    finally:
        if old is None:
            os.environ.pop("SN_S1_DEFER", None)
        else:
            os.environ["SN_S1_DEFER"] = old
    assert torch.equal(got, ref)
