"""GPU (-m gpu): the drop-in API as the fast path, the N > 1 branch of bench.py, config [4] at its real size.

  * `SchemaNetPredictor.forward` in eval() under no_grad captures the launch sequence behind the backbone by itself and
    replays it (reference entry: schema_inference/graph/__init__.py:37-57): replayed scores == eager scores bit for bit,
    a weight update re-captures, moving tap buffers fall back to eager launches;
  * `bench.py --gpus 2` under torch.distributed.run in rehearsal mode (both ranks on this one GPU over gloo): the code
    the driver runs for the scaling curve (reference scripts/init_schema_net.py:19-65, eval/evaluation.py:95-97);
  * three training iterations at config [4]'s real size (config/caltech_101/schema_net/deit_small-l9-M_1024.yaml:22-47:
    B = 64, M = 1024, K = 101, n_max = 1024, E = 256) against the plain-torch route of the same package.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import datagen
from test_gpu_parity import DEV, T, _Backbone, make_schema_net, mods, scores_close  # noqa: F401  (mods: fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, payload):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(payload, fh, indent=1)
    except OSError:
        pass


class _Rotating(torch.nn.Module):
    def __init__(self, batches):
        super().__init__()
        self.batches, self.i = batches, 0

    def forward(self, x):
        mid, ext = self.batches[self.i % len(self.batches)]
        self.i += 1
        return {"mid_feat": mid, "extracted": ext}


def _predictor(mods, batches, M, D, K, E, seed=0):
    graph = mods["graph"]
    disc = mods["disc"].Discretization(size=M, dim=D).to(DEV)
    with torch.no_grad():
        disc.vocabulary.weight.copy_(T(datagen.bellish((M, D), 303 + seed, 1.0)))
    wrapper = mods["Wrapper"](_Rotating(batches), mods["disc"].DiscretizationModule(disc)).to(DEV).eval()
    torch.manual_seed(seed)
    sn = make_schema_net(mods, M, K)
    sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)
    return graph.SchemaNetPredictor(wrapper, sn, m).eval(), wrapper


# =============================================================================== the API path replays itself
@pytest.mark.parametrize("E", [256, 32])
def test_predictor_replays_its_own_launch_sequence(mods, E):
    bs, H, L, D, M, K = 6, 3, 196, 192, 128, 5
    batches = [(T(datagen.bellish((L + 1, bs, D), 400 + i, 1.0)), T(datagen.bellish((bs * H, L + 1, L + 1), 410 + i, 2.0))) for i in range(3)]
    pred, wrapper = _predictor(mods, batches, M, D, K, E)
    x = torch.zeros(bs, 3, 4, 4, device=DEV)
    for cache in (True, False):                       # eval() default (class-graph features kept), and recomputed per call
        pred.matcher.cache_atlas = cache
        pred.matcher.invalidate_atlas_cache()
        pred.invalidate_graphs()
        pred.graph_replay = False
        wrapper.backbone_jit.i = 0
        with torch.no_grad():
            eager = [pred(x) for _ in range(3)]
        eager_pred = [o["pred"].clone() for o in eager]
        pred.graph_replay = True
        wrapper.backbone_jit.i = 0
        with torch.no_grad():
            first = [pred(x)["pred"] for _ in range(3)]                  # three captures (one per tap-buffer set)
            assert len(pred._graphs) == 3
            again = [pred(x) for _ in range(6)]                         # replays, twice around
        assert len(pred._graphs) == 3 and pred._graph_misses == 0        # (consecutive misses: a hit clears the count)
        assert all(len(e) == 2 for e in pred._graphs.values())          # one capture per tap-buffer set (the output ring is opt-in)
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(first[i], eager_pred[i]), f"capture pass differs from eager (cache {cache}, batch {i})"
            assert torch.equal(again[i]["pred"], eager_pred[i]) and torch.equal(again[3 + i]["pred"], eager_pred[i])
        assert list(again[0].keys()) == ["pred", "class_vertices", "class_edges", "class_ingredients"]
        # `class_edges` is computed when it is read (LazyOutputs): the key is there, iteration over keys does not compute,
        # reading gives the reference's tensor (golden G5 route: sn_atlas_normalize), the same from a replayed and an eager call
        assert "class_edges" in again[0] and len(again[0]) == 4 and again[0]._lazy
        ce = again[0]["class_edges"]
        assert not again[0]._lazy and torch.equal(ce, eager[0]["class_edges"]) and tuple(ce.shape) == (K, M, M)
        assert torch.equal(ce, pred.schema_net.get_atlas()["class_edges"])
        assert [tuple(v.shape) for v in again[1].values()][2] == (K, M, M)      # .values() / .items() compute as well
        assert again[0]["pred"].data_ptr() != again[3]["pred"].data_ptr()        # the caller owns what it gets
    # ---- a weight update: the captures are dropped for new ones, results follow the new weights
    with torch.no_grad():
        pred.matcher.gnn.layers[0].g_conv.linear.weight.mul_(1.25)      # bumps the version counter (as optimizer.step does)
        pred.schema_net.vertex_weights.tensor.mul_(0.5).add_(0.01)
        wrapper.backbone_jit.i = 0
        new = pred(x)["pred"]
        pred.graph_replay = False
        wrapper.backbone_jit.i = 0
        want = pred(x)["pred"]
        pred.graph_replay = True
    assert torch.equal(new, want) and not torch.equal(new, eager_pred[0])
    # ---- the reference's other uses stay eager: requires_graph (host-side list slicing), grad mode, train()
    with torch.no_grad():
        wrapper.backbone_jit.i = 0
        n_before = len(pred._graphs)
        full = pred(x, requires_graph=True)
        assert len(pred._graphs) == n_before and "instance_edges" in full
    scores_close(full["pred"], want, "requires_graph route vs replayed route")
    pred.train()
    assert len(pred._graphs) == 0


def test_predictor_output_ring_and_miss_counting(mods):
    """With `output_ring` (opt-in) `forward` hands out `pred` from a two-deep ring (no copy kernel between two graph
    launches): the tensor of call n is still intact after call n + 1 on the same taps and is rewritten by call n + 2; the
    default (`output_ring = False`) gives every call a tensor of its own.  A fine-tune loop that alternates weight updates and
    evaluation phases misses once per phase for ever: the miss count is of CONSECUTIVE misses, so replay stays on
    (a lifetime count switched it off silently after 32)."""
    bs, H, L, D, M, K = 4, 2, 196, 192, 128, 5
    mid, ext = T(datagen.bellish((L + 1, bs, D), 425, 1.0)), T(datagen.bellish((bs * H, L + 1, L + 1), 426, 2.0))
    mid2 = T(datagen.bellish((L + 1, bs, D), 427, 1.0))
    pred, wrapper = _predictor(mods, [(mid, ext)], M, D, K, 32)
    x = torch.zeros(bs, 3, 4, 4, device=DEV)
    assert pred.output_ring is False                      # default: the caller owns what `forward` returns (a copy), like the reference
    pred.output_ring = True
    with torch.no_grad():
        a = pred(x)["pred"]
        a0 = a.clone()
        mid.copy_(mid2)                                   # same buffers, new tokens: the next calls compute something else
        b = pred(x)["pred"]
        torch.cuda.synchronize()
        assert a.data_ptr() != b.data_ptr() and torch.equal(a, a0) and not torch.equal(a, b)
        c = pred(x)["pred"]                               # the ring comes round: call n + 2 rewrites call n's tensor
        torch.cuda.synchronize()
        assert c.data_ptr() == a.data_ptr() and torch.equal(c, b)
        pred.output_ring = False
        d, e = pred(x)["pred"], pred(x)["pred"]
        assert len({a.data_ptr(), b.data_ptr(), d.data_ptr(), e.data_ptr()}) == 4 and torch.equal(d, b) and torch.equal(e, b)
        pred.output_ring = True
        for phase in range(5 * pred.max_graphs):          # 40 phases: a weight update, then two evaluation calls
            pred.train()
            pred.matcher.gnn.layers[0].g_conv.linear.bias.add_(1.0e-3)
            pred.eval()
            pred(x), pred(x), pred(x)
        assert pred.graph_replay and pred._graph_misses == 0 and len(pred._graphs) == 1
        pred.graph_replay = False
        want = pred(x)["pred"]
        pred.graph_replay = True
        assert torch.equal(pred(x)["pred"], want)


def test_predictor_with_moving_taps_falls_back_to_eager(mods):
    """a backbone that returns NEW buffers on every call: captures cannot be reused; after 4 x max_graphs misses the
    predictor stops capturing (still correct, eager launches)"""
    bs, H, L, D, M, K = 2, 2, 196, 192, 128, 5
    mid, ext = T(datagen.bellish((L + 1, bs, D), 420, 1.0)), T(datagen.bellish((bs * H, L + 1, L + 1), 421, 2.0))
    keep = []

    class _Fresh(torch.nn.Module):
        def forward(self, x):
            keep.append((mid.clone(), ext.clone()))              # (kept alive: the allocator cannot hand the block out again)
            return {"mid_feat": keep[-1][0], "extracted": keep[-1][1]}
    pred, wrapper = _predictor(mods, [(mid, ext)], M, D, K, 32)
    wrapper.backbone_jit = _Fresh()
    pred.max_graphs = 2
    x = torch.zeros(bs, 3, 4, 4, device=DEV)
    with torch.no_grad():
        outs = [pred(x)["pred"].clone() for _ in range(11)]
    assert pred.graph_replay is False and len(pred._graphs) == 0
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_predictor_replay_with_a_backbone_that_allocates_its_outputs(mods):
    """A real backbone returns NEW tensors on every call and the caller drops them after the forward pass: the caching
    allocator then hands the next call the same blocks, and the predictor - which keeps no reference to the taps -
    replays the capture it made for those addresses (a reference would pin the blocks and every call would miss)."""
    bs, H, L, D, M, K = 4, 2, 196, 192, 128, 5
    mids = [T(datagen.bellish((L + 1, bs, D), 430 + i, 1.0)) for i in range(3)]
    exts = [T(datagen.bellish((bs * H, L + 1, L + 1), 440 + i, 2.0)) for i in range(3)]

    class _Allocating(torch.nn.Module):
        i = 0

        def forward(self, x):
            j = self.i % 3
            self.i += 1
            return {"mid_feat": mids[j] * 1.0, "extracted": exts[j] * 1.0}          # fresh outputs, nobody keeps them
    pred, wrapper = _predictor(mods, [(mids[0], exts[0])], M, D, K, 32)
    wrapper.backbone_jit = _Allocating()
    x = torch.zeros(bs, 3, 4, 4, device=DEV)
    with torch.no_grad():
        pred.graph_replay = False
        want = [pred(x)["pred"].clone() for _ in range(3)]
        pred.graph_replay = True
        wrapper.backbone_jit.i = 0
        got = [pred(x)["pred"] for _ in range(12)]
    torch.cuda.synchronize()
    for i, g_ in enumerate(got):
        assert torch.equal(g_, want[i % 3]), i
    assert pred.graph_replay and 1 <= len(pred._graphs) <= 3 and pred._graph_misses <= 3, (len(pred._graphs), pred._graph_misses)


def test_predict_batches_keeps_batches_in_flight_and_in_order(mods):
    """`SchemaNetPredictor.predict_batches`: an evaluation loop over 11 batches from an allocating backbone with three
    batches in flight on the predictor's own streams - every batch's `pred` equals the eager forward of the same batch,
    in the order of the loader; one capture per stream (never one capture on two streams); a second loop over the same
    predictor replays without a new capture; in train() / with autograd it degrades to one forward per batch."""
    bs, H, L, D, M, K = 4, 2, 196, 192, 128, 5
    n = 11
    xs = [torch.full((bs, 3, 2, 2), float(i), device=DEV) for i in range(n)]
    mids = [T(datagen.bellish((L + 1, bs, D), 450 + i, 1.0)) for i in range(n)]
    exts = [T(datagen.bellish((bs * H, L + 1, L + 1), 470 + i, 2.0)) for i in range(n)]

    class _Backbone(torch.nn.Module):                       # the batch is picked by the image tensor itself
        def forward(self, x):
            j = int(x[0, 0, 0, 0].item())
            return {"mid_feat": mids[j] * 1.0, "extracted": exts[j] * 1.0}
    pred, wrapper = _predictor(mods, [(mids[0], exts[0])], M, D, K, 32)
    wrapper.backbone_jit = _Backbone()
    with torch.no_grad():
        pred.graph_replay = False
        want = [pred(x)["pred"].clone() for x in xs]
        pred.graph_replay = True
        got = []
        for o in pred.predict_batches(iter(xs), depth=3):
            assert list(o.keys()) == ["pred", "class_vertices", "class_edges", "class_ingredients"]
            got.append(o["pred"] + 0.0)                      # (read on the caller's stream right away)
        torch.cuda.synchronize()
        assert len(got) == n
        for i in range(n):
            assert torch.equal(got[i], want[i]), i
        assert pred.graph_replay and 3 <= len(pred._graphs) <= pred.max_graphs
        assert len({k[-1] for k in pred._graphs}) == 3       # three streams, each with captures of its own
        misses = pred._graph_misses
        again = [o["pred"] for o in pred.predict_batches(xs, depth=3)]
        torch.cuda.synchronize()
        assert pred._graph_misses == misses
        for i in range(n):
            assert torch.equal(again[i], want[i]), i
        # tap buffers that keep moving: the replay gives up on the way and the loop goes on with eager launches on its streams
        pred.invalidate_graphs()
        pred._graph_misses = 4 * pred.max_graphs
        gave_up = [o["pred"] + 0.0 for o in pred.predict_batches(xs, depth=3)]
        torch.cuda.synchronize()
        assert pred.graph_replay is False
        for i in range(n):
            assert torch.equal(gave_up[i], want[i]), i
        pred.graph_replay = True
        pred._graph_misses = 0
    # not an inference loop: plain forwards, same results
    outs = [o["pred"] for o in pred.predict_batches(xs[:2], depth=3)]
    assert outs[0].requires_grad and torch.allclose(outs[0], want[0], rtol=1e-5, atol=1e-5 * float(want[0].abs().max()))


# =============================================================================== the N > 1 branch of bench.py
@pytest.mark.parametrize("world", [2, 3])
def test_bench_ranks_rehearsal(world):
    """What the driver launches for N > 1, with every rank on this box's one GPU over gloo (SN_BENCH_REHEARSAL=1: RCCL
    cannot form a multi-rank group on one device; the numbers mean nothing, the code path is the one of the scaling run):
    stdout is ONE JSON line, world_size as asked, every image of every region voted, the edge statistics (C2's
    26 214 500 floats: a multiple of 2, not of 3 or 8 - the zero-padded length) merged by reduce_scatter + all_gather.
    A GPU box admits six processes on its card - three ranks, the test runner and the launcher stay inside that (six ranks
    were killed by its process guard); world size 8 runs on the CPU over gloo:
    tests/test_host_cpu.py::test_statistics_eight_ranks_gloo_at_c2_length."""
    steps, port = (3 if world == 2 else 2), str(29600 + (os.getpid() + 17 * world) % 300)
    regions = 2 if world == 2 else 1
    env = dict(os.environ, SN_BENCH_REHEARSAL="1", SN_BENCH_BATCHES="4" if world == 2 else "2", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    if world > 2:
        env["SN_BENCH_DEPTH"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "2",
           "--regions", str(regions), "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["world_size"] == world and out["steps"] == steps and out["regions"] == regions
    assert out["votes_merged"] == 256 * steps * world
    assert out["config"]["global_batch"] == 256 * world and out["scaling"] == "weak"
    assert out["init_atlas"]["world_size"] == world and out["init_atlas"]["edge_stats_collective"] == "reduce_scatter+all_gather"
    assert out["value"] > 0 and abs(out["ms_per_step"] * steps * 1e-3 * out["value"] - 256 * world * steps) < 1e-3 * 256 * world * steps
    assert "REHEARSAL" in out["launch"] and out["cpu_baseline"] is None


def test_matcher_uses_the_padded_batch_behind_untouched_lists(mods):
    """`SchemaNet.forward` returns the reference's lists as views of its padded kernel output; `Matcher.forward` then takes
    that batch instead of padding 3 x bs tensors (and, in training, back-propagating through 3 x bs slices).  Same scores,
    same in-place padded lists and same gradients as the general route (lists the caller has touched), bit for bit."""
    graph = mods["graph"]
    B, L, M, K = 6, 196, 128, 5
    ing, attn, acls = datagen.graph_case(B, L, M, seed=77)
    torch.manual_seed(5)
    sn = make_schema_net(mods, M, K)
    sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
    m = graph.Matcher("inner_product", M, dict(embed_dim=256, num_layers=2, identity_proj=False, activation="relu")).to(DEV)

    def run(touch, train):
        sn.train(train); m.train(train)
        for p_ in list(sn.parameters()) + list(m.parameters()):
            p_.grad = None
        with torch.set_grad_enabled(train):
            inst = sn(T(ing), T(attn), T(acls))
            assert isinstance(inst, dict) and list(inst.keys()) == ["instance_ingredients", "instance_vertices", "instance_edges"]
            assert inst.padded is not None
            if touch:                                   # a caller that rebuilt a list entry: the general route
                inst["instance_vertices"][1] = inst["instance_vertices"][1] * 1.0
            pred = m(inst, sn.get_atlas())
            if train:
                pred.square().sum().backward()
        grads = [sn.vertex_attribute_weights.tensor.grad, sn.edge_attribute_weights.tensor.grad] if train else []
        return pred.detach(), inst, [g_.clone() for g_ in grads]

    for train in (False, True):
        fast, inst_f, g_f = run(False, train)
        slow, inst_s, g_s = run(True, train)
        assert torch.equal(fast, slow)
        n = max(len(x) for x in inst_f["instance_ingredients"])
        for k_ in ("instance_ingredients", "instance_vertices", "instance_edges"):
            for a_, b_ in zip(inst_f[k_], inst_s[k_]):
                assert a_.shape[0] == n and torch.equal(a_.detach(), b_.detach()), k_
        for a_, b_ in zip(g_f, g_s):
            assert (a_ - b_).abs().max().item() <= 1e-6 * b_.abs().max().item()


@pytest.mark.parametrize("rsl", [False, True])
def test_class_edges_with_fused_backward(mods, monkeypatch, rsl):
    """Training: `SchemaNet.get_class_edges()` as one HIP pass forward and one back (`ops.class_edges_autograd`) against the
    chain of torch ops of the reference (schema_net.py:152-175) differentiated by autograd: same class edges, same in-place
    pruning of the parameter, same gradient - zeros at clamped negatives, the gradient AT zero, NaN for every cell of a
    row whose sum is 0 (a pruned vertex), bit pattern of the NaNs aside."""
    graph = mods["graph"]
    K, n = 7, 96
    torch.manual_seed(21)
    vw = torch.rand(K, n)
    vw[:, ::5] = 0.0                                   # pruned vertices: their rows sum to 0
    ew = torch.randn(K, n, n)                          # negatives (clamped), and
    ew[:, 3, :] = 0.0                                  # an all-zero row of a kept vertex
    ew[1, 7, 9] = 0.0                                  # a single zero: clamp_min passes the gradient there
    gy = torch.randn(K, n, n)

    def run(fused):
        monkeypatch.setenv("SN_ATLAS_AUTOGRAD_FUSED", "1" if fused else "0")
        sn = graph.SchemaNet(num_vertices=n, num_classes=K, prune_node_threshold=0.001, remove_self_loop=rsl).to(DEV)
        with torch.no_grad():
            sn.vertex_weights.tensor.copy_(vw.to(DEV)); sn.edge_weights.tensor.copy_(ew.to(DEV))
        sn.train()
        ce = sn.get_class_edges()
        assert ce.requires_grad
        ce.backward(gy.to(DEV))
        return ce.detach().cpu(), sn.edge_weights.tensor.grad.cpu(), sn.edge_weights.tensor.detach().cpu()

    ce_f, g_f, p_f = run(True)
    ce_t, g_t, p_t = run(False)
    assert torch.equal(p_f, p_t)                        # the parameter, pruned in place
    assert torch.allclose(ce_f, ce_t, rtol=2e-6, atol=1e-9)
    assert torch.equal(torch.isnan(g_f), torch.isnan(g_t)) and torch.isnan(g_t).any() and not torch.isnan(g_t).all()
    ok = ~torch.isnan(g_t)
    assert torch.allclose(g_f[ok], g_t[ok], rtol=2e-6, atol=1e-9)
    assert float(g_t[ok].abs().max()) > 0


def test_linear_with_per_graph_weight_gradient(mods, monkeypatch):
    """Training route of the GCN with the library's GEMMs (SN_LINEAR_MFMA=0; the default since round 5 is the matrix-core form,
    tests/test_gpu_train_ops.py::test_linear_on_the_matrix_cores): the Linear of a layer takes its weight gradient as G per-graph
    products + one sum (the library's single [out, in] product over G n rows runs on 256 tiny tiles): same y, same three
    gradients as nn.Linear up to the order of the fp32 sums; taken on the GPU under autograd only."""
    from schema_inference.graph import gnn as gnn_mod
    monkeypatch.setenv("SN_LINEAR_MFMA", "0")
    torch.manual_seed(3)
    lin = torch.nn.Linear(96, 64).to(DEV)
    x = torch.randn(7, 50, 96, device=DEV, requires_grad=True)
    dy = torch.randn(7, 50, 64, device=DEV)
    y = gnn_mod._linear(lin, x)
    assert type(y.grad_fn).__name__.startswith("_LinearPerGraphWeightGrad")
    y.backward(dy)
    got = (y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None; lin.zero_grad()
    y2 = lin(x)
    y2.backward(dy)
    want = (y2.detach(), x.grad, lin.weight.grad, lin.bias.grad)
    for g_, w_ in zip(got, want):
        assert (g_ - w_).abs().max().item() <= 2e-6 * w_.abs().max().item()
    with torch.no_grad():
        assert gnn_mod._linear(lin, x).grad_fn is None
    assert not type(gnn_mod._linear(lin, x[0]).grad_fn).__name__.startswith("_LinearPerGraphWeightGrad")       # 2-D input: the library


# =============================================================================== the reference trainer's AMP route
def test_train_iter_under_autocast_with_grad_scaler(mods):
    """`use_amp: True` of the reference's trainer (worker_schema_net.py:128-143): forward + loss under `torch.autocast`, the
    scaled loss back-propagated, `GradScaler.step`.  The HIP-backed autograd.Functions of this package (`ops._SymAdjMatmul`,
    `_EdgesAdjMatmul`, `_ClassEdges`, `_RowEntropy`, `gnn._LinearPerGraphWeightGrad`) carry an autocast policy - inputs cast to
    fp32, autocast off inside - so the raw-pointer kernels never see half tensors: three iterations run, finite, and stay
    within fp16-matmul distance of the same three iterations without AMP."""
    from schema_inference import loss as loss_mod
    from schema_inference import train as train_mod
    graph = mods["graph"]
    B, L, M, K, E = 8, 196, 128, 5, 32
    g = lambda s_: torch.Generator().manual_seed(s_)  # noqa: E731
    ing = torch.randint(0, M, (B, L), generator=g(1))
    attn, acls = torch.randn(B, L, L, generator=g(2)), torch.randn(B, L, generator=g(3))
    label = torch.randint(0, K, (B,), generator=g(4))

    def run(amp):
        torch.manual_seed(11)
        sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(DEV)
        sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
        torch.manual_seed(12)
        m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)
        sn.train(); m.train()
        params = list(sn.parameters()) + list(m.parameters())
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-4)
        loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
        weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0) if amp else None
        seen = []

        def forward():
            inst = sn(ing.to(DEV), attn.to(DEV), acls.to(DEV))
            atlas = sn.get_atlas()
            out = {"pred": m(inst, atlas)}
            out.update(atlas)
            seen.append((torch.is_autocast_enabled(), out["class_edges"].dtype))
            return out
        losses = []
        for _ in range(3):
            total, _ = train_mod.train_iter(forward, sn, loss_fn, weights, opt, {"label": label.to(DEV)}, scaler=scaler)
            losses.append(float(total))
        assert all(a_ == amp for a_, _ in seen) and all(dt == torch.float32 for _, dt in seen)      # (the HIP class-edges op stays fp32 under autocast)
        return losses, {n_: p_.detach().clone() for n_, p_ in list(sn.named_parameters()) + list(m.named_parameters())}

    l_amp, p_amp = run(True)
    l_ref, p_ref = run(False)
    assert all(np.isfinite(l_amp)) and all(np.isfinite(l_ref))
    for a_, b_ in zip(l_amp, l_ref):
        assert abs(a_ - b_) <= 2e-2 * abs(b_), (l_amp, l_ref)
    for n_ in p_ref:
        pa, pr = p_amp[n_].nan_to_num(0), p_ref[n_].nan_to_num(0)
        assert torch.isfinite(p_amp[n_]).all() or not torch.isfinite(p_ref[n_]).all(), n_
        # (AdamW moves a parameter by ~lr per step whatever the size of its gradient: where a gradient is ~0 its sign, and so the
        # step, may differ between fp16 and fp32 matmuls - the two runs cannot be further apart than the three steps)
        # (the IR-Atlas parameters are also renormalised by schema_net.normalize() before every iteration, which scales such
        # differences: the bound is held for the GNN's weights, the atlas only has to stay finite)
        if n_.startswith("gnn."):
            assert float((pa - pr).abs().max()) <= 3.2e-3, n_


# =============================================================================== config [4] at its real size
def test_c5_real_size_training_iterations(mods):
    """deit_small-l9-M_1024.yaml:22-47: B = 64, M = 1024, K = 101, n_max = 1024 (a 424 MB edge_weights with gradients),
    E = 256.  Three `train_iter` steps (normalize -> forward -> SchemaInferenceLoss -> backward -> AdamW) with the
    adjacency products on the matrix cores (ops.edges_adj_matmul / sym_adj_matmul) against the same three steps with
    SN_GCN_MFMA=0 (library bmm: the plain-torch route of the same package) from the same initial state: losses within
    1e-4 relative, gradients of w_v, w_e, vertex_weights, edge_weights within 1e-3 of their scale; peak memory and step time are recorded."""
    from schema_inference import loss as loss_mod
    from schema_inference import train as train_mod
    graph = mods["graph"]
    B, L, D, M, K, E = 64, 196, 384, 1024, 101, 256
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    ing = torch.randint(0, M, (B, L), generator=g(1))
    ing[:, ::3] = ing[:, :1]                                                 # repeated words: graphs of ~130 vertices
    attn = torch.randn(B, L, L, generator=g(2))
    acls = torch.randn(B, L, generator=g(3))
    label = torch.randint(0, K, (B,), generator=g(4))

    def build():
        torch.manual_seed(11)
        sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(DEV)
        sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
        torch.manual_seed(12)
        m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)

        class _Model(torch.nn.Module):                                       # the part of SchemaNetPredictor behind the (frozen) wrapper
            def __init__(self):
                super().__init__()
                self.schema_net, self.matcher = sn, m

            def forward(self, batch):
                inst = self.schema_net(batch["ingredients"], batch["attn"].clone(), batch["attn_cls"].clone())
                atlas = self.schema_net.get_atlas()
                out = {"pred": self.matcher(inst, atlas)}
                out.update(atlas)
                return out
        return _Model()

    def run(mfma, graphed=False):
        old = os.environ.get("SN_GCN_MFMA")
        os.environ["SN_GCN_MFMA"] = "1" if mfma else "0"
        try:
            model = build().train()
            loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=5e-4, fused=True, capturable=graphed)
            if graphed:
                # the iteration as one hipGraph replay (train.GraphedTrainIter) over the route SchemaNetPredictor takes (padded batch)
                batch = {"ingredients": ing.to(DEV), "attn": attn.to(DEV), "attn_cls": acls.to(DEV)}
                target = {"label": label.to(DEV)}
                weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
                sn_, m_ = model.schema_net, model.matcher

                def padded(b):
                    atlas = sn_.get_atlas()
                    out = {"pred": m_.forward_padded(sn_.instance_graph_padded(b["ingredients"], b["attn"].clone(), b["attn_cls"].clone()), atlas)}
                    out.update(atlas)
                    return out
                if graphed == "eager":                                   # the same 2 + 6 steps as plain `train_iter` calls (on the route GraphedTrainIter takes)
                    sn_.compact_training = True
                    losses = [float(train_mod.train_iter(lambda: padded(batch), sn_, loss_fn, weights, opt, target)[0]) for _ in range(8)]
                    return losses[2:], None, None, None
                step = train_mod.GraphedTrainIter(padded, sn_, loss_fn, weights, opt, batch, target, warmup=2)
                assert step.memsets_replaced > 0 and step.memsets_left == 0     # (the library's multi-block reductions at this size)
                losses, times = [], []
                for it in range(6):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    total, _ = step(batch, target)
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - t0)
                    losses.append(float(total))
                return losses, times, None, torch.cuda.max_memory_allocated() / 2 ** 30
            batch = {"ingredients": ing.to(DEV), "attn": attn.to(DEV), "attn_cls": acls.to(DEV)}
            target = {"label": label.to(DEV)}
            weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            # gradients of the first iteration (train_iter clears them after its optimizer step): one forward / backward by hand
            opt.zero_grad(set_to_none=True)
            model.schema_net.normalize()
            train_mod.weighted_total(loss_fn(model(batch), target), weights).backward()
            sn = model.schema_net
            grads = {"w_v": sn.vertex_attribute_weights.tensor.grad.clone(), "w_e": sn.edge_attribute_weights.tensor.grad.clone(),
                     "edge_weights": sn.edge_weights.tensor.grad.clone(),
                     "vertex_weights": sn.vertex_weights.tensor.grad.clone()}
            losses, times = [], []
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                total, _ = train_mod.train_iter(lambda: model(batch), model.schema_net, loss_fn, weights, opt, target)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                losses.append(float(total))
            return losses, times, grads, torch.cuda.max_memory_allocated() / 2 ** 30
        finally:
            if old is None:
                os.environ.pop("SN_GCN_MFMA", None)
            else:
                os.environ["SN_GCN_MFMA"] = old

    l_mfma, t_mfma, g_mfma, mem_mfma = run(True)
    torch.cuda.empty_cache()
    l_lib, t_lib, g_lib, mem_lib = run(False)
    assert all(np.isfinite(l_mfma)) and all(np.isfinite(l_lib))
    for a, b in zip(l_mfma, l_lib):
        assert abs(a - b) <= 1e-4 * abs(b), (l_mfma, l_lib)
    for k in ("w_v", "w_e", "vertex_weights"):
        assert torch.isfinite(g_mfma[k]).all() and g_mfma[k].abs().max() > 0
        scale = g_lib[k].abs().max()
        assert (g_mfma[k] - g_lib[k]).abs().max() <= 1e-3 * scale, k
    # edge_weights: the rows of pruned vertices (vertex weight <= 0.001: about half of 1024 equal-ish weights) have a zero
    # row sum, and the reference's normalize_sum (utils.py:25-34: x / sum.detach(), nan_to_num) hands them 0 / 0 = NaN
    # gradients, which its normalize() (schema_net.py:133-142) turns into zeros before the next iteration; both routes
    # must agree on WHERE they are, be finite elsewhere and equal there
    ge, gl = g_mfma["edge_weights"], g_lib["edge_weights"]
    nan_e, nan_l = ~torch.isfinite(ge), ~torch.isfinite(gl)
    assert torch.equal(nan_e, nan_l)
    n_nan, n_all = int(nan_e.sum()), ge.numel()
    assert n_nan < n_all
    ge, gl = ge.masked_fill(nan_e, 0), gl.masked_fill(nan_l, 0)
    assert ge.abs().max() > 0 and (ge - gl).abs().max() <= 1e-3 * gl.abs().max()
    del ge, gl, g_mfma, g_lib
    torch.cuda.empty_cache()
    lg_mfma, tg_mfma, _, memg = run(True, graphed=True)
    torch.cuda.empty_cache()
    lg_lib, tg_lib, _, _ = run(False, graphed=True)
    assert all(np.isfinite(lg_mfma)) and all(np.isfinite(lg_lib))
    for i, (a, b) in enumerate(zip(lg_mfma, lg_lib)):            # (steps 3 .. 8 of the trajectory: the two routes drift apart slowly)
        assert abs(a - b) <= (1e-4 if i < 3 else 2e-3) * abs(b), (lg_mfma, lg_lib)
    # the replays follow the eager trajectory of the same route step for step (replay k = step k + 2: two warm-up steps).
    # With the graph's memset nodes left in place they did not: from the second replay on the library's multi-block
    # reductions kept the previous replay's result (12.18 instead of 12.07 at step 4).
    torch.cuda.empty_cache()
    le_mfma = run(True, graphed="eager")[0]
    np.testing.assert_allclose(lg_mfma, le_mfma, rtol=5e-6)
    _report("c5_real_size_training.json", {
        "shape": {"B": B, "M": M, "K": K, "n_max": M, "E": E, "edge_weights_MB": K * M * M * 4 / 2 ** 20},
        "edge_weight_gradients_nan_rows_of_pruned_vertices": n_nan / n_all,
        "optimizer": "torch.optim.AdamW(lr=1e-3, weight_decay=5e-4, fused=True)",
        "losses_mfma": l_mfma, "losses_library": l_lib, "iter_seconds_mfma": t_mfma, "iter_seconds_library": t_lib,
        "route_eager": "train_iter over the reference's python lists (SchemaNet.forward -> Matcher.forward: one host synchronisation)",
        "losses_mfma_graphed": lg_mfma, "losses_mfma_eager_same_route": le_mfma, "losses_library_graphed": lg_lib,
        "iter_seconds_mfma_graphed": tg_mfma, "iter_seconds_library_graphed": tg_lib,
        "route_graphed": "train.GraphedTrainIter over the padded batch (instance_graph_padded -> forward_padded), capturable AdamW: one hipGraph launch per iteration",
        "peak_GiB_mfma": mem_mfma, "peak_GiB_library": mem_lib, "peak_GiB_mfma_graphed": memg})


def test_c5_real_size_forward_matches_cpu_pipeline(mods):
    """Config [4] at its real size (B = 64, M = 1024, K = 101, class graphs of 1024 vertices: the 404 MB IR-Atlas) through
    the drop-in modules in eval() - `SchemaNet.forward` -> `get_atlas` -> `Matcher.forward` - against the reference's forward
    restated op for op on the host (oracle/cpu_pipeline.py: the reference's C++ graph stage, bmm / Linear / LayerNorm), run in
    fp32 AND in fp64: instance graphs equal, scores element-wise within 1e-5 and no further from fp64 than the reference's own
    fp32 (the training test at this size compares two routes of this package with each other)."""
    from oracle import cpu_pipeline
    from test_gpu_parity import as_good_as_fp32_reference
    graph = mods["graph"]
    B, L, M, K, E = 64, 196, 1024, 101, 256
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    ing = torch.randint(0, M, (B, L), generator=g(1))
    ing[:, ::3] = ing[:, :1]
    attn = torch.randn(B, L, L, generator=g(2))
    acls = torch.randn(B, L, generator=g(3))
    torch.manual_seed(11)
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001)
    sn.register_class_vertices(torch.stack([torch.randperm(M, generator=g(20 + k)) for k in range(K)]))
    with torch.no_grad():
        sn.vertex_weights.tensor[:, ::7] = 0.0                                   # pruned vertices
        sn.vertex_attribute_weights.tensor.copy_(torch.tensor([[0.3], [0.7]]))
        sn.edge_attribute_weights.tensor.copy_(torch.tensor([[0.6], [0.4]]))
    torch.manual_seed(12)
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    with torch.no_grad():
        for layer in m.gnn.layers:
            layer.norm.weight.uniform_(0.5, 1.5); layer.norm.bias.uniform_(-0.5, 0.5)
    P = {"gnn." + k: v.detach().clone() for k, v in m.gnn.state_dict().items()}
    with torch.no_grad():
        w_v, w_e = sn.vertex_attribute_weights.tensor.detach().clone(), sn.edge_attribute_weights.tensor.detach().clone()
        ids_c, v_c, e_c = cpu_pipeline.instance_graph(ing, attn.clone(), acls.clone(), w_v, w_e)
        cv, ce = cpu_pipeline.get_atlas(sn.vertex_weights.tensor.detach().clone(), sn.edge_weights.tensor.detach().clone())
        ci = sn.class_ingredients.tensor.clone()
        P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
        ref64 = cpu_pipeline.matcher(P64, [x.clone() for x in ids_c], [x.double() for x in v_c], [x.double() for x in e_c],
                                     cv.double(), ce.double(), ci, M)
        n_c = [len(x) for x in ids_c]
        ids_keep = [x.clone() for x in ids_c]
        ref32 = cpu_pipeline.matcher(P, ids_c, v_c, e_c, cv, ce, ci, M)
        sn, m = sn.to(DEV).eval(), m.to(DEV).eval()
        inst = sn(ing.to(DEV), attn.to(DEV), acls.to(DEV))
        atlas = sn.get_atlas()
        for b in (0, 17, B - 1):
            assert torch.equal(inst["instance_ingredients"][b].cpu(), ids_keep[b]) and len(inst["instance_ingredients"][b]) == n_c[b]
        assert torch.allclose(atlas["class_edges"].cpu(), ce, rtol=2e-6, atol=1e-9)
        got = m(inst, atlas).cpu()
    assert tuple(got.shape) == (B, K)
    rel = scores_close(got, ref32, "C5 real size forward")
    vs64 = as_good_as_fp32_reference(got, ref32, ref64, "C5 real size forward")
    _report("c5_real_size_forward.json", {"shape": {"B": B, "M": M, "K": K, "n_max": M, "E": E}, "max_rel_err_above_floor": rel, "vs_fp64": vs64})


# =============================================================================== a pruned IR-Atlas, compacted
@pytest.mark.parametrize("case", ["seventy_percent", "one_vertex_left_and_nothing_pruned"])
def test_compacted_class_graphs_of_a_pruned_atlas(mods, case):
    """A trained IR-Atlas is sparse: the loss's entropy terms push most vertices of a class under prune_node_threshold
    (reference schema_net.py:152-166), their rows and columns of the class graph are zeroed - isolated nodes.
    `get_atlas(fused_adjacency="compact")` builds the GCN operand from the kept vertices of each class only (per-class
    extents), `GNN` adds the isolated vertices' share of the class feature from a per-word table.  Against the reference's
    forward on the host in fp32 and fp64 (oracle/cpu_pipeline.py): scores within 1e-5 and no further from fp64 than the
    reference's own fp32; and against the uncompacted route of this package."""
    from oracle import cpu_pipeline
    from test_gpu_parity import as_good_as_fp32_reference
    graph = mods["graph"]
    K, M, E, B, L = 8, 512, 256, 5, 196
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(3)
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001)
    sn.register_class_vertices(torch.stack([torch.randperm(M, generator=g) for _ in range(K)]))
    with torch.no_grad():
        vw = torch.rand(K, M, generator=g)
        low = torch.rand(K, M, generator=g) < 0.7                           # ~70 % of every class under the threshold
        if case == "one_vertex_left_and_nothing_pruned":
            low[0] = True
            low[0, 17] = False                                              # ONE vertex kept: the class is its isolated nodes + one
            low[1] = False                                                  # nothing pruned (1 / 512 > threshold for equal weights)
            vw[1] = 0.5
        vw = torch.where(low, vw * 1.0e-4, vw)
        sn.vertex_weights.tensor.copy_(vw)
        sn.edge_weights.tensor.copy_(torch.rand(K, M, M, generator=g) * (torch.rand(K, M, M, generator=g) > 0.3))
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    with torch.no_grad():
        for layer in m.gnn.layers:
            layer.norm.weight.uniform_(0.5, 1.5); layer.norm.bias.uniform_(-0.5, 0.5)
    sizes = [L, 1, 77, 130, 150][:B]
    inst_ids = [torch.randperm(M, generator=g)[:s_].sort().values for s_ in sizes]
    inst_v = [torch.rand(s_, generator=g) for s_ in sizes]
    inst_e = []
    for s_ in sizes:
        e = torch.rand(s_, s_, generator=g) * (torch.rand(s_, s_, generator=g) > 0.5)
        inst_e.append((e / e.sum(-1, keepdim=True)).nan_to_num(0))
    P = {"gnn." + k: v.detach().clone() for k, v in m.gnn.state_dict().items()}
    with torch.no_grad():
        cv, ce = cpu_pipeline.get_atlas(sn.vertex_weights.tensor.detach().clone(), sn.edge_weights.tensor.detach().clone())
        ci = sn.class_ingredients.tensor.clone()
        frac = float((cv <= 0.001).float().mean())
        assert frac > 0.5 if case == "seventy_percent" else bool((cv[0] > 0.001).sum() == 1 and (cv[1] > 0.001).all())

        def host(dtype):
            Pd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
            return cpu_pipeline.matcher(Pd, [x.clone() for x in inst_ids], [x.to(dtype) for x in inst_v], [x.to(dtype) for x in inst_e],
                                        cv.to(dtype), ce.to(dtype), ci, M)
        ref32, ref64 = host(torch.float32), host(torch.float64)
        sn, m = sn.to(DEV).eval(), m.to(DEV).eval()
        inst = {"instance_ingredients": [x.to(DEV) for x in inst_ids], "instance_vertices": [x.to(DEV) for x in inst_v],
                "instance_edges": [x.to(DEV) for x in inst_e]}
        assert sn._atlas_compaction_pays()
        atlas_c = sn.get_atlas(fused_adjacency="compact")
        assert "class_perm" in atlas_c and atlas_c["class_n_kept"].tolist() == (cv > 0.001).sum(1).tolist()
        got_c = m(inst, atlas_c).cpu()
        # a second pass on the same parameter versions leaves the rows of the pruned vertices unread (zeroed in place by the
        # first): the same operand, the same scores
        assert sn._pruned_in_place is not None
        atlas_c2 = sn.get_atlas(fused_adjacency="compact")
        assert torch.equal(atlas_c2["class_n_kept"], atlas_c["class_n_kept"]) and torch.equal(atlas_c2["class_perm"], atlas_c["class_perm"])
        assert torch.equal(m(inst, atlas_c2).cpu(), got_c)                  # (blocks of the operand beyond a class's extent are never written: compare what is consumed)
        got_p = m(inst, sn.get_atlas(fused_adjacency=True)).cpu()
    scores_close(got_c, ref32, "compacted atlas vs the reference forward")
    vs64 = as_good_as_fp32_reference(got_c, ref32, ref64, "compacted atlas")
    scale = float(ref64.abs().max())
    assert float((got_c - got_p).abs().max()) <= 2e-6 * scale                # the two routes of this package: fp32 summation order apart
    _report(f"compacted_atlas_{case}.json", {"pruned_fraction": frac, "vs_fp64": vs64,
                                             "max_diff_compact_vs_plain_over_scale": float((got_c - got_p).abs().max()) / scale})


# =============================================================================== split-fp16 GCN outside its comfort zone
def _stress_case(name):
    """(K, n_cls, M, B, E, make_params(m), make_class_edges(g)) of one stress shape"""
    cases = {
        # IR-Atlas after sparsity training: 70 % of the weights exactly zero, the rest spanning 1e-8 .. 1 before the row normalisation
        "sparse_atlas": dict(K=8, n_cls=512, M=512, emb=1.0, gamma=(0.5, 1.5), sparse=True),
        # LayerNorm gamma up to 1e2, embeddings up to 1e3
        "large_weights": dict(K=6, n_cls=256, M=256, emb=1.0e3 / 2.0, gamma=(1.0, 1.0e2), sparse=False),
        # class graphs of 1024 vertices
        "n_max_1024": dict(K=4, n_cls=1024, M=1024, emb=1.0, gamma=(0.5, 1.5), sparse=True),
        # everything tiny: embeddings 1e-3, gamma 1e-2 (operands far below fp16's normal range when unscaled)
        "tiny_weights": dict(K=6, n_cls=256, M=256, emb=1.0e-3, gamma=(1.0e-3, 1.0e-2), sparse=False),
    }
    return cases[name]


@pytest.mark.parametrize("name", ["sparse_atlas", "large_weights", "n_max_1024", "tiny_weights"])
def test_split_fp16_gcn_is_as_good_as_the_fp32_reference(mods, name):
    """Matcher scores of the HIP path against the reference's forward (reference gnn.py:20-98, match.py:33-76; restated
    with its torch ops in oracle/cpu_pipeline.py) run twice on the host: in fp32 (what the reference computes) and in
    fp64 (the truth both approximate).  Criterion, element-wise:  |hip - fp64| <= |fp32 reference - fp64| + 1e-6 x score
    scale - the HIP path may not be further from the truth than the reference itself, up to a millionth of the scale.
    The operands leave the range an UNSCALED hi/lo fp16 split resolves (adjacency entries of 1e-8, features of 1e3 or
    1e-3): the power-of-two operand scales of csrc/sn_gcn.hip are what this test holds."""
    from oracle import cpu_pipeline
    graph = mods["graph"]
    c = _stress_case(name)
    K, n_cls, M, E, B, L = c["K"], c["n_cls"], c["M"], 256, 5, 196
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    torch.manual_seed(3)
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    with torch.no_grad():
        m.gnn.embedding.weight[:M].mul_(c["emb"])
        for layer in m.gnn.layers:
            layer.norm.weight.copy_(torch.empty(E).uniform_(*c["gamma"], generator=g) * (torch.randint(0, 2, (E,), generator=g) * 2 - 1))
            layer.norm.bias.copy_(torch.randn(E, generator=g) * c["gamma"][1] * 0.1)
    # ---- class graphs
    ew = torch.rand(K, n_cls, n_cls, generator=g)
    if c["sparse"]:
        ew = torch.pow(10.0, -8.0 * torch.rand(K, n_cls, n_cls, generator=g)) * (torch.rand(K, n_cls, n_cls, generator=g) > 0.7)
    ce = (ew / ew.sum(-1, keepdim=True)).nan_to_num(0)
    cv = torch.rand(K, n_cls, generator=g)
    cv = cv / cv.sum(-1, keepdim=True)
    ci = torch.stack([torch.randperm(M, generator=g)[:n_cls] for _ in range(K)])
    # ---- instance graphs (ragged, as Matcher pads them: match.py:48-54)
    sizes = [L, 1, 77, 130, 150][:B]
    inst_ids = [torch.randperm(M, generator=g)[:s].sort().values for s in sizes]
    inst_v = [torch.rand(s, generator=g) for s in sizes]
    inst_e = []
    for s in sizes:
        e = torch.rand(s, s, generator=g) * (torch.rand(s, s, generator=g) > 0.5)
        inst_e.append((e / e.sum(-1, keepdim=True)).nan_to_num(0))
    P = {"gnn." + k: v.detach().clone() for k, v in m.gnn.state_dict().items()}

    def host(dtype):
        Pd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
        return cpu_pipeline.matcher(Pd, [x.clone() for x in inst_ids], [x.to(dtype) for x in inst_v], [x.to(dtype) for x in inst_e],
                                    cv.to(dtype), ce.to(dtype), ci, M)
    with torch.no_grad():
        ref32, ref64 = host(torch.float32).double(), host(torch.float64)
        m = m.to(DEV)
        inst = {"instance_ingredients": [x.to(DEV) for x in inst_ids], "instance_vertices": [x.to(DEV) for x in inst_v],
                "instance_edges": [x.to(DEV) for x in inst_e]}
        got = m(inst, {"class_vertices": cv.to(DEV), "class_edges": ce.to(DEV), "class_ingredients": ci.to(DEV)}).cpu().double()
    assert torch.isfinite(got).all() and torch.isfinite(ref64).all()
    scale = ref64.abs().max().item()
    err_hip, err_ref = (got - ref64).abs(), (ref32 - ref64).abs()
    worst = (err_hip - err_ref).max().item()
    _report(f"gcn_stress_{name}.json", {"score_scale": scale, "max_err_hip_over_scale": err_hip.max().item() / scale,
                                        "max_err_fp32_reference_over_scale": err_ref.max().item() / scale,
                                        "max_excess_over_reference_error_over_scale": worst / scale})
    assert (err_hip <= err_ref + 1e-6 * scale).all(), (name, err_hip.max().item() / scale, err_ref.max().item() / scale, worst / scale)


# =============================================================================== dictionaries beyond the kernel's 256 output slots
def test_instance_e_with_a_dictionary_of_more_than_256_entries(mods):
    """`cpp_feat_to_instance_e` with the dictionary of a whole class graph (600 words -> slots, most of them absent from
    the image; one word of the image missing from it: slot 0, large_scale_feat_to_e.cpp:117-118): the reference has no
    size limit (:58-60), the kernel addresses 256 slots - the shim compacts and scatters.  Against the C oracle."""
    from oracle import cabi, pyops
    cx = mods["cx"]
    rng = np.random.default_rng(5)
    B, L, n_dict = 3, 196, 600
    ing = rng.integers(0, 700, (B, L)).astype(np.int64)
    attn = rng.normal(size=(B, L, L)).astype(np.float32)
    p_attn = np.exp(attn - attn.max(-1, keepdims=True))
    p_attn = (p_attn / p_attn.sum(-1, keepdims=True)).astype(np.float32)
    geo = pyops.pair_wise_point_sim(14, 14)
    w = np.asarray([[0.4], [0.6]], np.float32)
    dicts = []
    for b in range(B):
        pool = rng.permutation(700)
        keys = pool[pool != ing[b, 0]][:n_dict]         # a word of the image the dictionary lacks -> slot 0
        dicts.append({int(k): int(v) for k, v in zip(keys, rng.permutation(n_dict))})
    _, want = cabi.instance_e(ing, p_attn, geo, dicts, w, mean=True)
    got = cx.cpp_feat_to_instance_e(torch.from_numpy(ing), torch.from_numpy(p_attn), torch.from_numpy(geo), dicts, T(w), True, False)
    for g_, w_, d in zip(got, want, dicts):
        assert tuple(g_.shape) == (len(d), len(d))
        np.testing.assert_allclose(g_.cpu().numpy(), w_, rtol=5e-6, atol=1e-7)


def test_atlas_pass_writes_class_edges_as_a_by_product(mods):
    """get_atlas(fused_adjacency="with_edges") (what the predictor uses without autograd): the GCN operand is the one of the
    plain fused route bit for bit, and `class_edges` is what sn_atlas_normalize writes to within one rounding (w x (1 / row
    sum) against w / row sum; reference schema_net.py:152-175) - NaN / negative / pruned / infinite weights included."""
    ops = mods["ops"]
    g = torch.Generator().manual_seed(33)
    K, n = 5, 200
    vw = torch.rand(K, n, generator=g); vw[:, ::7] = 0.0
    ew = torch.randn(K, n, n, generator=g)
    ew[0, 3, 5] = float("nan"); ew[1, 4, :] = -1.0; ew[2, 6, 7] = float("inf")
    for rsl in (False, True):
        a, b, c = (ew.clone().to(DEV) for _ in range(3))
        cv0, ce0 = ops.atlas_normalize(vw.to(DEV), a, 0.004, rsl)
        cv1, adj1 = ops.atlas_adjacency_planes(vw.to(DEV), b, 0.004, rsl)
        cv2, adj2, ce2 = ops.atlas_adjacency_planes(vw.to(DEV), c, 0.004, rsl, want_edges=True)
        assert torch.equal(cv0, cv2) and torch.equal(cv1, cv2)
        assert torch.equal(a.nan_to_num(7.0), c.nan_to_num(7.0))                        # the same in-place pruning
        assert torch.equal(adj1.hi, adj2.hi) and torch.equal(adj1.lo, adj2.lo)
        assert torch.isfinite(ce2).all() and torch.equal(ce0 == 0, ce2 == 0)
        np.testing.assert_allclose(ce2.cpu().numpy(), ce0.cpu().numpy(), rtol=4e-7, atol=0)             # (two roundings instead of one: <= 2 ulp)


# =============================================================================== round 5: the instance side with one extent per graph
def test_adjacency_planes_per_graph_equal_the_masked_ones_where_they_are_read(mods):
    """`sn_gcn_adjacency_planes_per_graph` (reference operand: gnn.py:27-30 on the zero-padded batch, match.py:48-54): inside a graph's
    own extent (rows rounded up to 32, k to 16) the planes are those of the masked producer bit for bit; what lies beyond is not
    produced (the buffer keeps its poison)."""
    ops = mods["ops"]
    G, n = 9, 196
    g = torch.Generator().manual_seed(12)
    e = torch.rand(G, n, n, generator=g).to(DEV)
    nv = torch.tensor([1, 15, 16, 33, 113, 128, 129, 160, 196], dtype=torch.int32, device=DEV)
    ext = nv.max().reshape(1).to(torch.int32)
    ref = ops.gcn_adjacency_planes(e, extent=ext, n_valid=nv)
    got = ops.gcn_adjacency_planes(e, extent=ext, n_valid=nv, per_graph=True)
    d_ref, d_got = ref.to_dense()[1], got.to_dense()[1]                  # [G, rows padded to 32, k padded to 16]
    for i, c in enumerate(nv.tolist()):
        rows, k = (c + 31) // 32 * 32, (c + 15) // 16 * 16
        assert torch.equal(d_got[i, :rows, :k], d_ref[i, :rows, :k]), (i, c)


@pytest.mark.parametrize("E", [256])
def test_instance_gnn_with_per_graph_extents(mods, monkeypatch, E):
    """GNN.forward without autograd on instance graphs with their own vertex counts (1 .. 196: below, at and above the 128-row
    tile): extents per graph (the default since round 5) against the batch maximum for everybody (SN_GCN_GRAPH_EXTENTS=0) and
    against the float64 forward on the host - the pooled graph features, element-wise."""
    import copy
    from schema_inference.graph import gnn as gnn_mod
    M, n = 512, 196
    torch.manual_seed(21)
    net = gnn_mod.GNN(M, E, 2).to(DEV).eval()
    g = torch.Generator().manual_seed(6)
    counts = [1, 2, 15, 16, 17, 100, 113, 127, 128, 129, 130, 160, 196, 64, 31, 32]
    G = len(counts)
    nv = torch.tensor(counts, dtype=torch.int32)
    mask = torch.arange(n)[None, :] >= nv[:, None]
    nodes = torch.rand(G, n, generator=g); nodes[mask] = 0
    edges = torch.rand(G, n, n, generator=g) / n
    edges = edges * (~mask)[:, :, None] * (~mask)[:, None, :]
    ids = torch.randint(0, M, (G, n), generator=g); ids[mask] = M
    n_max = nv.max().reshape(1).to(torch.int32)
    with torch.no_grad():
        want = copy.deepcopy(net).double().cpu()(nodes.double(), edges.double(), ids, feat_mask=mask, divisor=n_max)
        run = lambda: net(nodes.to(DEV), edges.to(DEV), ids.to(DEV), n_valid=nv.to(DEV), divisor=n_max.to(DEV)).double().cpu()      # noqa: E731
        got = run()
        monkeypatch.setenv("SN_GCN_GRAPH_EXTENTS", "0")
        ref = run()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= float((ref - want).abs().max()) + 1e-6 * scale
    assert float((got - ref).abs().max()) <= 2e-6 * scale


def test_matcher_copies_and_pickles_after_it_has_run(mods):
    """`copy.deepcopy` (an EMA copy of the model) and `pickle` of a Matcher that has already run - in inference (class branch on a
    side stream, cached handle) and in training (instance pass on a second stream): the per-process stream objects and the cached
    handle stay behind, the copy computes the same scores."""
    import copy
    import pickle
    graph = mods["graph"]
    M, K, E, B, L = 128, 5, 256, 4, 196
    torch.manual_seed(2)
    sn = make_schema_net(mods, M, K)
    sn.register_class_vertices(torch.arange(M, device=DEV).repeat(K, 1))
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)
    g = torch.Generator().manual_seed(3)
    ing = torch.randint(0, M, (B, L), generator=g).to(DEV)
    attn, acls = torch.randn(B, L, L, generator=g).to(DEV), torch.randn(B, L, generator=g).to(DEV)

    def scores(matcher, train):
        matcher.train(train); sn.train(train)
        with torch.set_grad_enabled(train):
            gr = sn.instance_graph_padded(ing, attn.clone(), acls.clone())
            if train:
                return matcher.forward_padded(gr, sn.get_atlas()).detach().clone()
            h = matcher.atlas_features_async(lambda: sn.get_atlas(detach=True, fused_adjacency=True))
            return matcher.forward_padded(gr, None, feat_kg=h).clone()
    want_eval, want_train = scores(m, False), scores(m, True)
    assert getattr(m, "_side_stream", None) is not None or getattr(m, "_train_stream", None) is not None
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert getattr(clone, "_side_stream", None) is None and getattr(clone, "_train_stream", None) is None
        assert torch.equal(scores(clone, False), want_eval)
        got = scores(clone, True)
        assert float((got - want_train).abs().max()) <= 1e-6 * float(want_train.abs().max())
