"""GPU tests of the training-side streaming passes of round 4 (csrc/sn_train.hip) and of the summed edge gradient of the
GCN layers: each against the chain of torch ops it replaces.  Reference ops: `SchemaNet.normalize`
schema_inference/graph/schema_net.py:133-142 (graph/utils.py:7-13), the GCN operand schema_inference/graph/gnn.py:27-31 and
its autograd chain rule."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "schemanet-pytorch_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from test_gpu_parity import DEV, mods  # noqa: E402,F401  (mods: fixture)

pytestmark = pytest.mark.gpu


def _torch_pow2_scale(x, top=8192.0):
    b = x.abs().amax().to(torch.float32).reshape(1)
    s = torch.exp2(torch.floor(torch.log2(top / b)).clamp(-60.0, 60.0))
    return torch.where(torch.isfinite(s) & (b > 0), s, torch.ones_like(s))


@pytest.mark.parametrize("shape", [(7,), (3, 1000), (101, 1024, 256), (5, 333, 17)])
def test_pow2_scale_kernel(mods, shape):
    """one read of the operand + one finishing launch == abs / amax / log2 / floor / exp2 of the library: the scale is the
    power of two that puts the largest magnitude into (2^12, 2^13]; zeros, NaN, inf give 1; tiny / huge magnitudes clamp"""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(DEV)
    for mul in (1.0, 1e-7, 3.3e5, 1e-30, 1e30, 2.0 ** -13, 0.0):
        xs = x * mul
        s = ops.pow2_scale(xs)
        assert s.shape == (1,) and s.dtype == torch.float32
        sv, amax = float(s), float(xs.abs().max())
        if amax == 0.0:
            assert sv == 1.0
            continue
        assert np.log2(sv) == np.floor(np.log2(sv)) and 2.0 ** -60 <= sv <= 2.0 ** 60
        if 2.0 ** -47 < amax < 2.0 ** 73:
            assert 4096.0 < sv * amax <= 8192.0, (mul, sv, amax)
        # the library chain agrees except when 8192 / amax is within an ulp of a power of two (its log2 rounds up there)
        want = float(_torch_pow2_scale(xs))
        assert sv == want or sv == want / 2, (mul, sv, want)
    xs = x.clone(); xs.view(-1)[xs.numel() // 2] = float("nan")
    assert float(ops.pow2_scale(xs)) == 1.0
    xs = x.clone(); xs.view(-1)[0] = float("inf")
    assert float(ops.pow2_scale(xs)) == 1.0
    exact = torch.zeros(shape, device=DEV); exact.view(-1)[-1] = -2.0        # 8192 / 2 = 2^12 exactly, negative element, last position
    assert float(ops.pow2_scale(exact)) == 4096.0


@pytest.mark.parametrize("G,n", [(1, 1), (3, 63), (2, 64), (3, 100), (2, 130), (2, 1024)])
def test_sym_half_inplace(mods, G, n):
    ops = mods["ops"]
    g = torch.Generator().manual_seed(G * 100 + n)
    s = torch.randn(G, n, n, generator=g).to(DEV)
    want = (s + s.transpose(1, 2)) * 0.5
    got = ops.sym_half_(s.clone())
    assert torch.equal(got, want)


@pytest.mark.parametrize("K,n,diag", [(3, 1024, True), (2, 1024, False), (5, 100, True), (2, 4100, True), (4, 513, False), (3, 8192, False)])
def test_normalize_sum_rows(mods, K, n, diag):
    """`SchemaNet.normalize()` on one parameter as one pass: clamp_min, row sum, divide, NaN -> 0, diagonal -> 0; rows of
    NaN (what AdamW leaves in the rows of pruned vertices, whose gradient is 0 / 0) and all-zero rows become zeros"""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(K * 10 + n)
    shape = (K, n, n) if diag else (K * 7, n)
    x = (torch.rand(*shape, generator=g) - 0.2).to(DEV)                      # some negatives: clamped
    x2 = x.view(-1, n)
    x2[1] = float("nan")
    x2[2] = 0.0
    x2[3, 5] = float("nan")
    x2[4] = -1.0                                                             # clamps to an all-zero row: 0 / 0 -> 0
    x2[5, 0] = float("inf")
    want = x.clone().clamp_min_(0.0)
    want.div_(want.sum(dim=-1, keepdim=True)).nan_to_num_(0)
    if diag:
        want.diagonal(dim1=1, dim2=2).fill_(0)
    got = ops.normalize_sum_rows_(x.clone(), 0.0, zero_diagonal=diag)
    assert torch.isfinite(got).all()
    assert torch.equal(got == 0, want == 0)
    assert (got - want).abs().max().item() <= 4e-7 * want.abs().max().item()            # (the row sum is taken in another order)
    rows = got.view(-1, n)
    assert float(rows[1].abs().max()) == 0.0 and float(rows[2].abs().max()) == 0.0 and float(rows[4].abs().max()) == 0.0
    # min_val > 0 (vertex weights use 0 too, but the argument is the reference's)
    got2 = ops.normalize_sum_rows_(x.clone(), 0.25)
    want2 = x.clone().clamp_min_(0.25)
    want2.div_(want2.sum(dim=-1, keepdim=True)).nan_to_num_(0)
    assert (got2 - want2).abs().max().item() <= 4e-7 * want2.abs().max().item()


def test_schema_net_normalize_takes_the_fused_pass_and_bumps_the_version(mods):
    graph = mods["graph"]
    torch.manual_seed(5)
    sn = graph.SchemaNet(num_vertices=1024, num_classes=3, prune_node_threshold=0.001).to(DEV)
    with torch.no_grad():
        sn.edge_weights.tensor.uniform_(-0.1, 1.0)
        sn.edge_weights.tensor[1, 7] = float("nan")
        sn.vertex_weights.tensor.uniform_(-0.1, 1.0)
    ref_e = sn.edge_weights.tensor.detach().clone().clamp_min_(0)
    ref_e.div_(ref_e.sum(-1, keepdim=True)).nan_to_num_(0)
    if sn.remove_self_loop:
        ref_e.diagonal(dim1=1, dim2=2).fill_(0)
    v0 = sn.edge_weights.tensor._version
    sn.normalize()
    assert sn.edge_weights.tensor._version > v0
    got = sn.edge_weights.tensor.detach()
    assert (got - ref_e).abs().max().item() <= 4e-7 * ref_e.abs().max().item() and float(got[1, 7].abs().max()) == 0.0
    old = os.environ.get("SN_NORMALIZE_FUSED")
    os.environ["SN_NORMALIZE_FUSED"] = "0"
    try:
        with torch.no_grad():
            sn.edge_weights.tensor.uniform_(-0.1, 1.0)
        ref = sn.edge_weights.tensor.detach().clone()
        sn.normalize()
        lib_route = sn.edge_weights.tensor.detach().clone()
    finally:
        os.environ.pop("SN_NORMALIZE_FUSED") if old is None else os.environ.__setitem__("SN_NORMALIZE_FUSED", old)
    with torch.no_grad():
        sn.edge_weights.tensor.copy_(ref)
    sn.normalize()
    assert (sn.edge_weights.tensor.detach() - lib_route).abs().max().item() <= 4e-7 * lib_route.abs().max().item()


@pytest.mark.parametrize("G,m,n,k", [(3, 200, 200, 256), (2, 1024, 1024, 64), (1, 70, 33, 48)])
def test_gemm_accumulates_into_c(mods, G, m, n, k):
    ops = mods["ops"]
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(G, m, k, generator=g).to(DEV)
    b = torch.randn(G, n, k, generator=g).to(DEV)
    c0 = torch.randn(G, m, n, generator=g).to(DEV)
    ap, bp = ops.split_planes(a), ops.split_planes(b)
    prod = ops.gcn_gemm(ap, bp, G, want_c=True)["c"]
    c = c0.clone()
    out = ops.gcn_gemm(ap, bp, G, accumulate_into=c)["c"]
    assert out.data_ptr() == c.data_ptr()
    assert torch.equal(c, c0 + prod)


@pytest.mark.parametrize("G,n,E", [(3, 196, 256), (2, 300, 64)])
def test_summed_edge_gradient_of_a_gnn(mods, G, n, E):
    """two GraphConv-like layers in sequence on one edge tensor: with sum_edge_grads the edge gradient reaches autograd once,
    as sym(S1 + S2), and equals the float64 chain rule of (E + E^T)/2 + I; a third consumer of the edges (the loss's entropy
    term) still adds its own gradient; the per-layer form (sum_edge_grads False) gives the same values"""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(G * 31 + n)
    e = torch.rand(G, n, n, generator=g)
    x = torch.randn(G, n, E, generator=g)
    w = torch.randn(E, E, generator=g) / E ** 0.5
    dy = torch.randn(G, n, E, generator=g) * 1e-4

    def run(dtype, dev, prod):
        ed = e.to(dev, dtype).requires_grad_(True)
        xd = x.to(dev, dtype).requires_grad_(True)
        wd = w.to(dev, dtype)
        h1 = torch.relu(prod(ed, xd) @ wd)
        h2 = prod(ed, h1) @ wd
        extra = (ed * ed).sum() * 1e-6                               # another consumer of the same tensor
        (h2 * dy.to(dev, dtype)).sum().add(extra).backward()
        return ed.grad.detach().double().cpu(), xd.grad.detach().double().cpu()

    def prod64(ed, xd):
        return torch.bmm((ed + ed.transpose(1, 2)) / 2 + torch.eye(n, dtype=ed.dtype), xd)

    want_e, want_x = run(torch.float64, "cpu", prod64)
    for summed in (True, False):
        planes = {}

        def prod(ed, xd):
            if "p" not in planes:
                planes["p"] = ops.gcn_adjacency_planes(ed.detach())
            return ops.edges_adj_matmul(ed, xd, planes["p"], sum_edge_grads=summed)
        got_e, got_x = run(torch.float32, DEV, prod)
        assert planes["p"].pending == 0 and planes["p"].grad_sum is None
        for got, want, what in ((got_e, want_e, "d edges"), (got_x, want_x, "d x")):
            err = (got - want).abs().max().item()
            assert err <= 4e-6 * want.abs().max().item() * max(1.0, (n / 196) ** 0.5), (summed, what, err, want.abs().max().item())


def test_graphed_train_iter_follows_the_eager_trajectory(mods):
    """`train.GraphedTrainIter` (normalize -> forward -> loss -> backward -> AdamW as one hipGraph replay) against `train_iter`
    called the same number of times from the same state with the same (capturable) optimizer: the losses of every step and
    the parameters at the end agree; batches of different content go through the static buffers."""
    from schema_inference import loss as loss_mod
    from schema_inference import train as train_mod
    graph = mods["graph"]
    B, L, M, K, E = 8, 49, 512, 4, 64
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    batches = []
    for i in range(3):
        ing = torch.randint(0, M, (B, L), generator=g(10 + i)); ing[:, ::3] = ing[:, :1]
        batches.append(({"ingredients": ing.to(DEV), "attn": torch.randn(B, L, L, generator=g(20 + i)).to(DEV),
                         "attn_cls": torch.randn(B, L, generator=g(30 + i)).to(DEV)},
                        {"label": torch.randint(0, K, (B,), generator=g(40 + i)).to(DEV)}))

    def build():
        torch.manual_seed(3)
        sn = graph.SchemaNet(num_vertices=M, num_classes=K, feat_h=7, feat_w=7, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0,
                             prune_node_threshold=0.001).to(DEV)
        sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
        torch.manual_seed(4)
        m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)

        class _Model(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.schema_net, self.matcher = sn, m

            def forward(self, batch):                                       # the route SchemaNetPredictor takes behind its wrapper
                atlas = self.schema_net.get_atlas()
                g_ = self.schema_net.instance_graph_padded(batch["ingredients"], batch["attn"].clone(), batch["attn_cls"].clone())
                out = {"pred": self.matcher.forward_padded(g_, atlas)}
                out.update(atlas)
                return out
        model = _Model().train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=5e-4, capturable=True, fused=True)
        return model, opt

    loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
    weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
    warm, steps = 2, 5
    model_e, opt_e = build()
    losses_e = []
    for i in range(warm):
        train_mod.train_iter(lambda: model_e(batches[0][0]), model_e.schema_net, loss_fn, weights, opt_e, batches[0][1])
    for i in range(steps):
        b, t = batches[i % 3]
        total, _ = train_mod.train_iter(lambda: model_e(b), model_e.schema_net, loss_fn, weights, opt_e, t)
        losses_e.append(float(total))
    model_g, opt_g = build()
    step = train_mod.GraphedTrainIter(model_g, model_g.schema_net, loss_fn, weights, opt_g, batches[0][0], batches[0][1], warmup=warm)
    losses_g = []
    for i in range(steps):
        b, t = batches[i % 3]
        total, ld = step(b, t)
        losses_g.append(float(total))
        assert set(ld) == {"cls", "entropy_vertex", "entropy_edge", "re_entropy_vertex", "re_entropy_edge"}
    assert np.isfinite(losses_g).all() and len(set(losses_g)) > 1
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-6)
    for (name, pe), (_, pg) in zip(model_e.named_parameters(), model_g.named_parameters()):
        pe, pg = pe.detach(), pg.detach()
        assert torch.equal(torch.isfinite(pe), torch.isfinite(pg)), name
        pe, pg = pe.nan_to_num(0), pg.nan_to_num(0)
        assert (pe - pg).abs().max().item() <= 2e-6 * max(1e-30, pe.abs().max().item()), name
    with pytest.raises(ValueError):
        train_mod.GraphedTrainIter(model_e, model_e.schema_net, loss_fn, weights, torch.optim.AdamW(model_e.parameters()), *batches[0])


def test_rectify_linear_select_form(mods):
    """the GPU form of the loss's rectifier (a select instead of a python branch on a device scalar): values and gradients of
    the reference's expression on both sides of `a` and at the pole of the unselected branch"""
    from schema_inference import loss as loss_mod
    for a in (0.0, 3.0):
        for xv in (a - 2.0, a - 1e-3, a, a + 1e-3, a + 1.0, a + 5.0):
            xc = torch.tensor(xv, dtype=torch.float32, requires_grad=True)
            xd = torch.tensor(xv, dtype=torch.float32, device=DEV, requires_grad=True)
            yc, yd = loss_mod.rectify_linear(xc, a), loss_mod.rectify_linear(xd, a)
            yc.backward(); yd.backward()
            assert float(yc.detach()) == pytest.approx(float(yd.detach()), rel=1e-6, abs=1e-7), (a, xv)
            assert torch.isfinite(xd.grad) and float(xc.grad) == pytest.approx(float(xd.grad), rel=1e-6), (a, xv)


@pytest.mark.parametrize("K,n,rsl", [(3, 1024, True), (4, 512, False), (2, 100, True), (3, 130, False)])
def test_class_edges_with_row_entropies(mods, K, n, rsl):
    """`class_edges_autograd(with_entropy=True)`: class_edges and, from the same pass, its row entropies; backwards one pass for
    both upstream gradients.  Against the two separate ops (sn_atlas_normalize + sn_row_entropy and their backward passes joined
    by autograd's add): the same bits forward, the same bits backward - also with only one of the two outputs used, with NaN
    rows (a class vertex whose whole row is pruned: 0 / 0) and with a zero upstream entropy gradient in all rows but one."""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(K * 7 + n)
    vw = torch.rand(K, n, generator=g).to(DEV)
    vw[:, ::3] *= 1e-4                                                      # a third of the vertices under the threshold
    ew0 = (torch.rand(K, n, n, generator=g) - 0.1).to(DEV)
    w_ce = torch.randn(K, n, n, generator=g).to(DEV) * 1e-3
    thr = 0.5 / n

    def run(fused, use_ce=True, use_ent=True):
        ew = ew0.clone().requires_grad_(True)
        ce = ops.class_edges_autograd(ew, vw, thr, rsl, with_entropy=fused)
        ent = getattr(ce, "_sn_row_entropy", (None, None))[1] if fused else ops.row_entropy(ce, 1.0e-7)
        loss = 0.0
        if use_ce:
            loss = loss + (ce * w_ce).sum()
        if use_ent:
            loss = loss + ent.max(dim=1)[0].mean() * 0.75                  # the loss's form: a maximum over the rows of a class
        loss.backward()
        return ce.detach(), ent.detach(), ew.grad.detach()

    for use_ce, use_ent in ((True, True), (False, True), (True, False)):
        ce_a, ent_a, g_a = run(False, use_ce, use_ent)
        ce_b, ent_b, g_b = run(True, use_ce, use_ent)
        assert torch.equal(ce_a, ce_b)
        assert torch.equal(ent_a, ent_b)
        assert torch.equal(torch.isnan(g_a), torch.isnan(g_b)) and bool(torch.isnan(g_a).any())
        assert torch.equal(g_a.nan_to_num(0), g_b.nan_to_num(0)), (use_ce, use_ent, (g_a.nan_to_num(0) - g_b.nan_to_num(0)).abs().max().item())
    assert ent_b.shape == (K, n) and float(ent_b.max()) > 0


def test_loss_takes_the_precomputed_row_entropies(mods):
    from schema_inference import loss as loss_mod
    graph = mods["graph"]
    torch.manual_seed(9)
    sn = graph.SchemaNet(num_vertices=256, num_classes=3, prune_node_threshold=0.001).to(DEV)
    atlas = sn.get_atlas()
    ce = atlas["class_edges"]
    assert ce.requires_grad and hasattr(ce, "_sn_row_entropy")
    ent = loss_mod.entropy(ce)
    assert ent is ce._sn_row_entropy[1]
    assert torch.equal(ent.detach(), mods["ops"].row_entropy(ce.detach(), 1.0e-7))
    assert loss_mod.entropy(ce, eps=1e-5) is not ce._sn_row_entropy[1]     # another eps: computed
    with torch.no_grad():
        assert not hasattr(sn.get_atlas()["class_edges"], "_sn_row_entropy")


@pytest.mark.parametrize("G,n,E,relu,masked", [(3, 196, 256, True, True), (101, 64, 256, True, False), (2, 70, 1024, True, True), (4, 33, 48, False, True),
                                               (1, 1, 64, True, False), (5, 130, 512, False, False)])
def test_mask_layernorm_act_with_autograd(mods, G, n, E, relu, masked):
    """the GNN layer's tail as one differentiable op: the forward values are those of the inference kernel (bit for bit) and of
    masked_fill + LayerNorm + ReLU in float64; dx, d gamma, d beta against float64 autograd of that chain, padded rows
    included (no gradient into them, their share of d beta kept)"""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(G * 17 + n + E)
    x = torch.randn(G, n, E, generator=g) * (0.5 + 2.0 * torch.rand(G, n, 1, generator=g))
    gamma = 0.5 + torch.rand(E, generator=g)
    beta = 0.3 * torch.randn(E, generator=g)
    dy = torch.randn(G, n, E, generator=g) * 1e-3
    n_valid = torch.randint(max(1, n // 2), n + 1, (G,), generator=g).to(torch.int32) if masked else None
    xd = x.to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    nv = None if n_valid is None else n_valid.to(DEV)
    y = ops.mask_layernorm_act(xd, gd, bd, 1e-5, n_valid=nv, relu=relu)
    y.backward(dy.to(DEV))
    inplace = ops.mask_layernorm_act_(x.to(DEV).clone(), gamma.to(DEV), beta.to(DEV), 1e-5, n_valid=nv, relu=relu)
    assert torch.equal(y.detach(), inplace)
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    h = x64
    if n_valid is not None:
        h = h.masked_fill((torch.arange(n)[None, :] >= n_valid[:, None])[..., None], 0)
    y64 = torch.nn.functional.layer_norm(h, (E,), g64, b64, 1e-5)
    if relu:
        y64 = torch.relu(y64)
    y64.backward(dy.double())
    for got, want, what, tol in ((y.detach(), y64.detach(), "y", 3e-6), (xd.grad, x64.grad, "dx", 2e-5), (gd.grad, g64.grad, "d gamma", 2e-5),
                                 (bd.grad, b64.grad, "d beta", 2e-5)):
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= tol * max(1e-30, want.abs().max().item()), (what, err, want.abs().max().item())
    if n_valid is not None:
        pad = (torch.arange(n)[None, :] >= n_valid[:, None]).to(DEV)
        assert float(xd.grad[pad].abs().max()) == 0.0 if bool(pad.any()) else True


@pytest.mark.parametrize("rows,E,shape", [(1025, 256, (101, 1024)), (129, 64, (5, 300)), (33, 512, (2, 7)), (17, 48, (3, 40))])
def test_embedding_backward_from_a_cached_sort(mods, rows, E, shape):
    """table[ids] with the sort of `ids` taken once (ops.sorted_ids_of): the lookup equals F.embedding, the table gradient equals
    its autograd (float64 sums), the padding row gets none, words that do not occur get zeros"""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(rows + E)
    w = torch.randn(rows, E, generator=g)
    ids = torch.randint(0, rows, shape, generator=g)
    ids[0, 0] = rows - 1                                             # the padding index occurs
    ids[ids == 3] = 4                                                # ... and word 3 never does
    dy = torch.randn(*shape, E, generator=g)
    wd = w.to(DEV).requires_grad_(True)
    idd = ids.to(DEV)
    order, seg = ops.sorted_ids_of(idd, rows)
    y = ops.embedding_sorted(wd, idd, order, seg, padding_idx=rows - 1)
    y.backward(dy.to(DEV))
    w64 = w.double().requires_grad_(True)
    y64 = torch.nn.functional.embedding(ids, w64, padding_idx=rows - 1)
    y64.backward(dy.double())
    assert torch.equal(y.detach().cpu(), y64.detach().float())
    err = (wd.grad.double().cpu() - w64.grad).abs().max().item()
    assert err <= 2e-6 * w64.grad.abs().max().item(), err
    assert float(wd.grad[rows - 1].abs().max()) == 0.0 and float(wd.grad[3].abs().max()) == 0.0


def test_gnn_uses_the_cached_sort_for_buffer_ids_only(mods):
    graph = mods["graph"]
    torch.manual_seed(2)
    gnn = graph.GNN(num_codes=64, embed_dim=32, num_layers=2).to(DEV).train()
    ids_buf = torch.nn.Parameter(torch.randint(0, 64, (3, 40), device=DEV), requires_grad=False)
    ids_plain = ids_buf.detach().clone()
    nodes = torch.rand(3, 40, device=DEV)
    edges = torch.rand(3, 40, 40, device=DEV)
    outs = []
    for ids in (ids_buf, ids_plain):
        gnn.zero_grad()
        out = gnn(nodes, edges, ids)
        out.sum().backward()
        outs.append((out.detach().clone(), gnn.embedding.weight.grad.detach().clone()))
    assert hasattr(ids_buf, "_sn_sorted") and not hasattr(ids_plain, "_sn_sorted")
    assert torch.equal(outs[0][0], outs[1][0])
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 2e-6 * outs[1][1].abs().max().item()
    v = ids_buf._sn_sorted[0]
    with torch.no_grad():
        ids_buf.copy_(torch.randint(0, 64, (3, 40), device=DEV))     # new content: the version moves, the sort is taken again
    gnn(nodes, edges, ids_buf)
    assert ids_buf._sn_sorted[0] != v


@pytest.mark.parametrize("G,n,E,with_div", [(3, 196, 256, True), (101, 64, 64, False), (2, 33, 48, True)])
def test_weighted_pool_with_autograd(mods, G, n, E, with_div):
    """the node-weighted mean pooling (reference gnn.py:96) as one differentiable op: value and both gradients against float64"""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(G + n + E)
    feat = torch.randn(G, n, E, generator=g)
    nodes = torch.rand(G, n, generator=g)
    dy = torch.randn(G, E, generator=g)
    div = torch.tensor([n - 3], dtype=torch.int32) if with_div else None
    fd, nd = feat.to(DEV).requires_grad_(True), nodes.to(DEV).requires_grad_(True)
    out = ops.weighted_pool_autograd(fd, nd, None if div is None else div.to(DEV))
    out.backward(dy.to(DEV))
    f64, n64 = feat.double().requires_grad_(True), nodes.double().requires_grad_(True)
    o64 = (f64 * n64[..., None]).sum(1) / (float(div) if div is not None else n)
    o64.backward(dy.double())
    for got, want, what in ((out.detach(), o64.detach(), "pooled"), (fd.grad, f64.grad, "d feat"), (nd.grad, n64.grad, "d nodes")):
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= 3e-6 * want.abs().max().item(), (what, err)


def test_graphed_train_iter_over_the_predictor(mods):
    """the documented use: `GraphedTrainIter(lambda batch: predictor(batch["x"]), ...)` over a `SchemaNetPredictor` in train() -
    backbone taps (no grad) -> S1 -> instance graphs -> atlas -> matcher, the reference's output dictionary into
    SchemaInferenceLoss - is capturable as it is (no host synchronisation on that route) and follows the eager iterations"""
    import datagen
    from schema_inference import loss as loss_mod
    from schema_inference import train as train_mod
    from test_gpu_api import _predictor
    from test_gpu_parity import T
    bs, H, L, D, M, K, E = 6, 3, 196, 192, 512, 5, 64
    mid, ext = T(datagen.bellish((L + 1, bs, D), 500, 1.0)), T(datagen.bellish((bs * H, L + 1, L + 1), 501, 2.0))
    x = torch.zeros(bs, 3, 4, 4, device=DEV)
    target = {"label": torch.randint(0, K, (bs,), generator=torch.Generator().manual_seed(7)).to(DEV)}
    loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
    weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}

    def run(graphed):
        pred, _ = _predictor(mods, [(mid, ext)], M, D, K, E, seed=3)
        pred.train()
        params = [p for n, p in pred.named_parameters() if p.requires_grad and not n.startswith("ingredient_wrapper")]
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-4, fused=True, capturable=True)
        fwd = lambda b: pred(b["x"])                                        # noqa: E731
        losses = []
        if graphed:
            step = train_mod.GraphedTrainIter(fwd, pred.schema_net, loss_fn, weights, opt, {"x": x}, target, warmup=2)
            for _ in range(4):
                losses.append(float(step({"x": x}, target)[0]))
        else:
            for i in range(6):
                total, _ = train_mod.train_iter(lambda: fwd({"x": x}), pred.schema_net, loss_fn, weights, opt, target)
                if i >= 2:
                    losses.append(float(total))
        return losses
    eager, graphed = run(False), run(True)
    assert np.isfinite(graphed).all() and len(set(graphed)) > 1
    np.testing.assert_allclose(graphed, eager, rtol=2e-6)


# =============================================================================== round 5: the Linear layers of training on the matrix cores
def _close_to_f64(got, want64, ref32, what, slack=1e-6):
    """|hip - f64| <= |fp32 library - f64| + slack * scale, element-wise maxima (the bar of DESIGN 4)"""
    scale = want64.abs().max().item()
    err, ref = (got.double().cpu() - want64).abs().max().item(), (ref32.double().cpu() - want64).abs().max().item()
    assert err <= ref + slack * scale, (what, err, ref, scale)


@pytest.mark.parametrize("shape,out_f,bias", [((7, 50, 96), 64, True), ((5, 130, 256), 256, True), ((101, 256), 256, True), ((3, 64, 32), 48, False)])
def test_linear_on_the_matrix_cores(mods, shape, out_f, bias):
    """`ops.linear_mfma` (reference gnn.py:31, the Linear of a graph convolution, and gnn.py:98, fc): y and the three gradients
    against float64, no further from it than the library's fp32 Linear; gradients of the size training sees (1e-4)."""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(sum(shape) + out_f)
    x = torch.randn(*shape, generator=g)
    w = torch.randn(out_f, shape[-1], generator=g) / shape[-1] ** 0.5
    b = torch.randn(out_f, generator=g) if bias else None
    dy = torch.randn(*shape[:-1], out_f, generator=g) * 1e-4

    def run(dtype, dev, fn):
        xd, wd = x.to(dev, dtype).requires_grad_(True), w.to(dev, dtype).requires_grad_(True)
        bd = b.to(dev, dtype).requires_grad_(True) if bias else None
        y = fn(xd, wd, bd)
        y.backward(dy.to(dev, dtype))
        return [y.detach(), xd.grad, wd.grad] + ([bd.grad] if bias else [])
    want = [t.double().cpu() for t in run(torch.float64, "cpu", torch.nn.functional.linear)]
    ref = run(torch.float32, DEV, torch.nn.functional.linear)
    got = run(torch.float32, DEV, ops.linear_mfma)
    for g_, w_, r_, what in zip(got, want, ref, ("y", "dx", "dw", "db")):
        assert g_.shape == w_.shape
        _close_to_f64(g_, w_, r_, what)


def test_linear_routes_of_the_gcn(mods, monkeypatch):
    """`gnn._linear`: under autograd on the GPU the matrix-core form (3-D and 2-D inputs); SN_LINEAR_MFMA=0: the library's GEMMs with
    the per-graph weight gradient (3-D) / nn.Linear (2-D); without autograd nn.Linear"""
    from schema_inference.graph import gnn as gnn_mod
    torch.manual_seed(3)
    lin = torch.nn.Linear(96, 64).to(DEV)
    x = torch.randn(7, 50, 96, device=DEV, requires_grad=True)
    assert type(gnn_mod._linear(lin, x).grad_fn).__name__.startswith("_LinearMfma")
    assert type(gnn_mod._linear(lin, x[0]).grad_fn).__name__.startswith("_LinearMfma")
    with torch.no_grad():
        assert gnn_mod._linear(lin, x).grad_fn is None
    monkeypatch.setenv("SN_LINEAR_MFMA", "0")
    assert type(gnn_mod._linear(lin, x).grad_fn).__name__.startswith("_LinearPerGraphWeightGrad")
    assert not type(gnn_mod._linear(lin, x[0]).grad_fn).__name__.startswith(("_LinearPerGraphWeightGrad", "_LinearMfma"))


@pytest.mark.parametrize("G,n,rows,F,cached", [(3, 196, 513, 256, False), (4, 128, 101, 64, True), (2, 300, 1025, 256, True)])
def test_gather_adj_matmul_with_autograd(mods, G, n, rows, F, cached):
    """`ops.gather_adj_matmul` = ((E + E^T)/2 + I) @ table[ids] + b (layer 1 of the GCN with its Linear folded into the table,
    reference gnn.py:27-31 re-associated): the product and the gradients of the edges, the table and the bias against float64;
    the padding row of the table gets no gradient; with and without a cached sort of the ids."""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(G * 7 + n + rows)
    e = torch.rand(G, n, n, generator=g) / n
    tab = torch.randn(rows, F, generator=g)
    b = torch.randn(F, generator=g)
    ids = torch.randint(0, rows, (G, n), generator=g)
    ids[:, -5:] = rows - 1                                           # padding positions
    dy = torch.randn(G, n, F, generator=g) * 1e-4

    def run64():
        ed, td, bd = (t.double().requires_grad_(True) for t in (e, tab, b))
        adj = (ed + ed.transpose(1, 2)) / 2 + torch.eye(n, dtype=torch.float64)
        y = torch.bmm(adj, torch.nn.functional.embedding(ids, td, padding_idx=rows - 1)) + bd
        y.backward(dy.double())
        return y.detach(), ed.grad, td.grad, bd.grad

    def run32(fn):
        ed, td, bd = (t.to(DEV).requires_grad_(True) for t in (e, tab, b))
        y = fn(ed, td, bd)
        y.backward(dy.to(DEV))
        return y.detach(), ed.grad, td.grad, bd.grad
    ids_d = ids.to(DEV)

    def lib(ed, td, bd):
        adj = (ed + ed.transpose(1, 2)) / 2 + torch.eye(n, device=DEV)
        return torch.bmm(adj, torch.nn.functional.embedding(ids_d, td, padding_idx=rows - 1)) + bd
    sort = ops.sorted_ids_of(ids_d, rows) if cached else None
    want, ref = run64(), run32(lib)
    got = run32(lambda ed, td, bd: ops.gather_adj_matmul(ed, td, ids_d, bd, None, sort, rows - 1, sum_edge_grads=False))
    for g_, w_, r_, what in zip(got, want, ref, ("y", "d edges", "d table", "d bias")):
        _close_to_f64(g_, w_, r_, what, slack=2e-6)
    assert float(got[2][rows - 1].abs().max()) == 0.0


@pytest.mark.parametrize("G,n,M,E", [(3, 196, 512, 256), (5, 64, 100, 64)])
def test_gnn_training_route_with_the_folded_first_layer(mods, monkeypatch, G, n, M, E):
    """GNN.forward under autograd: layer 1 folded into the embedding table + every Linear on the matrix cores (default) against
    the route of round 4 (SN_TRAIN_FOLD=0, SN_LINEAR_MFMA=0: fp32 embedding, library GEMMs) and against float64 on the host: the graph
    features and the gradients of every parameter, of the edges and of the node weights."""
    from schema_inference.graph import gnn as gnn_mod
    torch.manual_seed(11)
    net = gnn_mod.GNN(M, E, 2).to(DEV)
    g = torch.Generator().manual_seed(5)
    nodes = torch.rand(G, n, generator=g)
    edges = torch.rand(G, n, n, generator=g) / n
    ids = torch.randint(0, M, (G, n), generator=g)
    n_valid = torch.tensor([n, n - 7, n // 2, n - 1, 3][:G], dtype=torch.int32)
    mask = torch.arange(n)[None, :] >= n_valid[:, None]
    ids[mask] = M
    nodes[mask] = 0
    edges = edges * (~mask)[:, :, None] * (~mask)[:, None, :]
    dout = torch.randn(G, E, generator=g) * 1e-3

    def run(model, dev, dtype):
        model.zero_grad()
        nd, ed = nodes.to(dev, dtype).requires_grad_(True), edges.to(dev, dtype).requires_grad_(True)
        out = model(nd, ed, ids.to(dev), feat_mask=mask.to(dev))
        out.backward(dout.to(dev, dtype))
        return [out.detach(), nd.grad, ed.grad] + [p.grad.clone() for p in model.parameters()]
    import copy
    net64 = copy.deepcopy(net).double().cpu()
    want = [t.double() for t in run(net64, "cpu", torch.float64)]
    got = run(net, DEV, torch.float32)
    monkeypatch.setenv("SN_TRAIN_FOLD", "0")
    monkeypatch.setenv("SN_LINEAR_MFMA", "0")
    ref = run(net, DEV, torch.float32)
    names = ["out", "d nodes", "d edges"] + [k for k, _ in net.named_parameters()]
    for g_, w_, r_, what in zip(got, want, ref, names):
        _close_to_f64(g_, w_, r_, what, slack=3e-6)


# =============================================================================== round 5: training with compacted class graphs
def _pruned_atlas(mods, K, M, frac_pruned, seed):
    """SchemaNet whose classes have `frac_pruned` of their vertex weights far under the prune threshold (and the atlas pruned accordingly
    by the first forward pass, as in a trained model)"""
    torch.manual_seed(seed)
    sn = mods["graph"].SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(DEV)
    sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        vw = torch.rand(K, M, generator=g) + 0.5
        drop = torch.rand(K, M, generator=g) < abs(frac_pruned)
        vw[drop] = 1.0e-7
        if frac_pruned < 0:
            vw[0] = 1.0                                              # M > 1000 equal weights: 1 / M < the threshold - a class with NO kept vertex
        sn.vertex_weights.tensor.copy_(vw.to(DEV))
        sn.edge_weights.tensor.copy_(torch.rand(K, M, M, generator=g).to(DEV))
    return sn


@pytest.mark.parametrize("K,M,E,frac", [(5, 256, 256, 0.5), (3, 200, 64, 0.8), (4, 128, 256, 0.0), (2, 1024, 64, -0.5)])
def test_training_with_compacted_class_graphs(mods, monkeypatch, K, M, E, frac):
    """Matcher.atlas_features under autograd on a pruned IR-Atlas: the class GNN on the kept vertices of every class + the per-word share of
    the pruned ones (SN_TRAIN_COMPACT=2: always) against the uncompacted route (=0) - the class features and the gradients of every
    GNN parameter, of the vertex weights and of the edge weights (NaN rows of pruned vertices at the same places, reference
    schema_net.py:152-175) - and both against float64 where the reference is finite."""
    graph = mods["graph"]
    sn = _pruned_atlas(mods, K, M, frac, 31 + K)
    torch.manual_seed(5)
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV).train()
    sn.train()
    dout = torch.randn(K, E, generator=torch.Generator().manual_seed(9)).to(DEV) * 1e-2

    def run(mode):
        monkeypatch.setenv("SN_TRAIN_COMPACT", mode)
        m.zero_grad(); sn.zero_grad()
        atlas = sn.get_atlas()
        assert ("class_perm" in atlas) == (mode == "2")
        feat = m.atlas_features(atlas)
        feat.backward(dout)
        grads = [p.grad.detach().clone() for p in m.gnn.parameters()]
        return feat.detach().clone(), grads, sn.vertex_weights.tensor.grad.detach().clone(), sn.edge_weights.tensor.grad.detach().clone()
    f0, g0, gv0, ge0 = run("0")
    f1, g1, gv1, ge1 = run("2")
    scale = float(f0.abs().max())
    assert float((f1 - f0).abs().max()) <= 3e-6 * scale, (float((f1 - f0).abs().max()), scale)
    for a, b, (name, _) in zip(g1, g0, m.gnn.named_parameters()):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, (name, float((a - b).abs().max()), float(b.abs().max()))
    assert float((gv1 - gv0).abs().max()) <= 2e-5 * float(gv0.abs().max()) + 1e-12
    nan0, nan1 = torch.isnan(ge0), torch.isnan(ge1)
    assert torch.equal(nan0, nan1)
    fin = ~nan0
    assert float((ge1[fin] - ge0[fin]).abs().max()) <= 2e-5 * float(ge0[fin].abs().max()) + 1e-12


@pytest.mark.parametrize("rows,E,n_ids", [(1025, 256, 64 * 196), (101, 64, 1000), (513, 1024, 300), (7, 16, 0)])
def test_embedding_backward_without_a_sort(mods, rows, E, n_ids):
    """`sn_embedding_grad_scan` (reference gnn.py:83, the gradient of nn.Embedding for ids that change with every batch): every row of
    the table's gradient against float64, no further from it than the library's embedding backward; the padding row zero; twice the
    same bits (a fixed summation order)."""
    from cpp_extension import _native as N
    lib = mods["cx"].load()
    g = torch.Generator().manual_seed(rows + E)
    ids = torch.randint(0, rows, (n_ids,), generator=g)
    if n_ids > 10:
        ids[:7] = rows - 1                                           # the padding row
        ids[7:40] = 3                                                # a word with many occurrences
    dy = torch.randn(n_ids, E, generator=g) * 1e-3
    ids_d, dy_d = ids.to(DEV), dy.to(DEV)

    def run():
        out = torch.full((rows, E), float("nan"), device=DEV)
        N.check(lib.sn_embedding_grad_scan(N.ptr(dy_d), N.ptr(ids_d), n_ids, rows, E, rows - 1, N.ptr(out), N.stream_ptr(torch.device(DEV, 0))), "scan")
        return out
    got = run()
    assert torch.equal(got, run())
    want = torch.zeros(rows, E, dtype=torch.float64).index_add_(0, ids, dy.double())
    want[rows - 1] = 0
    ref = torch.ops.aten.embedding_dense_backward(dy_d, ids_d, rows, rows - 1, False) if n_ids else torch.zeros(rows, E, device=DEV)
    scale = float(want.abs().max()) if n_ids else 1.0
    assert float((got.double().cpu() - want).abs().max()) <= float((ref.double().cpu() - want).abs().max()) + 1e-6 * scale
    assert float(got[rows - 1].abs().max()) == 0.0


def test_operand_splits_with_node_extents_and_corner_scatter(mods):
    """The helpers of the compacted training route against plain torch: `split_planes(..., node_extents=)` equals the plain split
    inside every graph's extent (rows rounded up to 32 in the plain form, nodes to the k chunk in the transposed one) bit for bit;
    `sym_scatter_corner` = the symmetrised corner scattered through the permutation, zero where a pruned vertex is involved
    (reference gnn.py:27-30: the chain rule through (E + E^T)/2 + I); a product with `zero_c`, `rows_valid` and `m_extent` leaves no
    element of its fp32 result undefined."""
    ops = mods["ops"]
    G, n, E = 5, 200, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn(G, n, E, generator=g).to(DEV)
    ext = torch.tensor([0, 1, 33, 128, 200], dtype=torch.int32, device=DEV)
    full, part = ops.split_planes(x, scale=None), ops.split_planes(x, scale=None, node_extents=ext)
    d_full, d_part = full.to_dense()[1], part.to_dense()[1]
    for i, c in enumerate(ext.tolist()):
        r = (c + 31) // 32 * 32
        assert torch.equal(d_part[i, :r], d_full[i, :r]), i
    full_t, part_t = ops.split_planes(x, scale=None, transpose=True), ops.split_planes(x, scale=None, transpose=True, node_extents=ext)
    dt_full, dt_part = full_t.to_dense()[1], part_t.to_dense()[1]                       # [G, E, nodes padded to 16]
    for i, c in enumerate(ext.tolist()):
        assert torch.equal(dt_part[i, :, :c], dt_full[i, :, :c]), i
    # ---- the corner gradient
    perm = torch.stack([torch.randperm(n, generator=g) for _ in range(G)]).to(torch.int32).to(DEV)
    corner = torch.randn(G, n, n, generator=g).to(DEV)
    want = torch.zeros(G, n, n, dtype=torch.float64)
    c64, p = corner.double().cpu(), perm.long().cpu()
    for i, k in enumerate(ext.tolist()):
        sym = (c64[i, :k, :k] + c64[i, :k, :k].t()) / 2
        want[i][p[i, :k][:, None], p[i, :k][None, :]] = sym
    got = ops.sym_scatter_corner(corner.clone(), perm, ext)
    assert float((got.double().cpu() - want).abs().max()) <= 1.2e-7 * float(want.abs().max())       # (one fp32 rounding of the half sum)
    assert torch.equal(got == 0, (want == 0).to(DEV))
    # ---- no undefined element behind the extents
    a, b = ops.split_planes(x, scale=None, node_extents=ext), ops.split_planes(torch.randn(96, E, generator=g).to(DEV), scale=None)
    bias = torch.randn(96, generator=g).to(DEV)
    c = ops.gcn_gemm(a, b, G, bias=bias, want_c=True, zero_c=True, m_extent=ext, rows_valid=ext)["c"]
    assert bool(torch.isfinite(c).all())
    ref = torch.matmul(x, b.to_dense()[0][0].t()) + bias
    for i, k in enumerate(ext.tolist()):
        assert float(c[i, k:].abs().max()) == 0.0 if k < n else True
        if k:
            assert float((c[i, :k] - ref[i, :k]).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 196, 196), (8, 196), (3, 5, 7), (1,), (0, 4)])
def test_weigh_attributes_fused(mods, shape, monkeypatch):
    """`attr2 @ w` behind the instance graphs under autograd (reference large_scale_feat_to_e.cpp:141-147, schema_net.py:302-305) as one
    HIP pass each way (round 6): the forward values are those of the torch form `a0 * w0 + a1 * w1` bit for bit, the gradient of the two
    weights is the float64 sum to fp32 rounding and no further from it than the torch form's."""
    from cpp_extension import ops
    g = torch.Generator().manual_seed(len(shape) * 100 + sum(shape))
    attr2 = torch.rand(*shape, 2, generator=g).to(DEV)
    w0 = torch.tensor([[0.35], [0.65]])
    grad = torch.randn(*shape, generator=g).to(DEV)

    def run(fused):
        monkeypatch.setenv("SN_WEIGH_FUSED", "1" if fused else "0")
        w = w0.clone().to(DEV).requires_grad_(True)
        out = ops.weigh_attributes(attr2, w)
        out.backward(grad)
        return out.detach(), w.grad.detach()
    out_f, dw_f = run(True)
    out_t, dw_t = run(False)
    assert tuple(out_f.shape) == tuple(shape) and tuple(dw_f.shape) == (2, 1)
    assert torch.equal(out_f, out_t)
    want = (grad.double().unsqueeze(-1) * attr2.double()).reshape(-1, 2).sum(0).reshape(2, 1)
    scale = (grad.double().abs().unsqueeze(-1) * attr2.double()).reshape(-1, 2).sum(0).reshape(2, 1) + 1e-30
    err_f = ((dw_f.double() - want).abs() / scale).max().item() if attr2.numel() else 0.0
    err_t = ((dw_t.double() - want).abs() / scale).max().item() if attr2.numel() else 0.0
    assert err_f <= 1e-6 and err_f <= err_t + 2e-7, (err_f, err_t)
