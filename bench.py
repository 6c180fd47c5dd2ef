#!/usr/bin/env python3
"""Schema-inference throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch that is already resident in HBM:
  tokens [B,197,384] f32 --S1 assign--> ingredients [B,196]
  head-averaged attention logits [B,197,197] --S2+S3 instance graph (cls slicing, clamp,
  softmax, grouping, normalise fused)--> padded instance graphs
  IR-Atlas normalise (K=100, n_max=512) --S4--> GCN on instances and on the atlas --> pred [B,100]
The class-graph branch (atlas normalise + GCN over the K class graphs) is recomputed in every step, like the reference
(`feat_kg_cache: off`).  What DOES survive across steps are operands derived from weights only, which are constant in
inference: the packed fp16 image of the codebook (S1) and `GNN.prepare()` (the layer-1 Linear folded into the embedding
table - the reference runs linear(embedding) in every forward -, the fp16 planes of W2, fc^T); `config.caches_across_steps`
names them.
The timed region is repeated: `--regions R` (default 7) back-to-back regions of exactly K steps, each bracketed by a
barrier + synchronize on both sides (max over ranks per region); `value` / `ms_per_step` are the MEDIAN region,
`value_min` / `value_max` the slowest / fastest (one region of 20 replays is 6 ms: clock state, not a measurement).
The K timed steps replay captured hipGraphs of the step (schema_inference.utils.graph_replay: same kernels,
no host launch path in the timed region).  There are SN_BENCH_BATCHES (default 8) DIFFERENT input batches resident in
HBM (8 x 117 MB: more than the 256 MB Infinity Cache), one capture per batch with its own buffers, visited in
rotation; SN_BENCH_DEPTH (default 4) of them are in flight on as many streams = `value`; the same K steps replayed
one at a time (depth 1) are reported as `value_depth1` (captured with the class branch forked onto a second stream, which
is what fills the chip when ONE step runs at a time; the pipelined captures run every step in line on its own stream, the
better choice with several in flight - see Matcher.atlas_features_async; that leg runs first: right after the captures, which are
host work with an idle GPU, so the timed region of a short run does not start on idle clocks).  A further, untimed pass of K eager steps with HIP events on
the launch stream gives the per-kernel durations of the roofline figures (`SN_BENCH_EAGER=1` times the eager loop
instead).
Workload = BASELINE.json configs[1]: DeiT-Small + CIFAR-100, B=256 per GPU, 512-word codebook.
Multi-GPU: images are sharded over ranks (weak scaling, B per rank fixed), no data-path
collective; the per-class prediction histogram + (n_seen) are all-reduced once at the end of
the timed region over RCCL (the eval-meter merge of the reference, eval/evaluation.py:95-97).
`value_api` (N = 1): images/sec through the drop-in API itself - `SchemaNetPredictor.forward(x)` called once per batch on the
RAW backbone taps (sequence-first tokens [197,256,384], per-head attention logits [256*6,197,197]: the fused head-mean
input of SURVEY 8(d)), one call at a time, no hand-written step (the predictor captures and replays the launch sequence
behind the backbone by itself); `value_api_batches`: the same calls made by `SchemaNetPredictor.predict_batches` (the
predictor's own evaluation loop, four batches in flight); `c1_value` / `c4_value` / `c5_value`: the same step at the shapes of configs[0], [3] and [4] (inference).
Outside the timed region every run also times one IR-Atlas initialisation over a synthetic image shard per rank
(`init_atlas`): with N > 1 its two merges are the RCCL collectives of the per-class schema statistics
(reference scripts/init_schema_net.py:19-65; 105 MB of edge sums at this configuration).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "schemanet-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
B, L, D, M, K, E, H = 256, 196, 384, 512, 100, 256, 6


def make_codebook(device):
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    pool = torch.randn(4096, D, generator=g(1))
    return (pool[torch.randperm(4096, generator=g(11))[:M]] + 0.05 * torch.randn(M, D, generator=g(2))).to(device)


def make_batch(rank, device, batch=0):
    """seeded CPU generators, then copied (same bits on every box).  Every rank and every batch
    index gets its own images (seed offset); batch 0 of rank 0 is the batch of SURVEY 8(d) (seeds 0 and 3)."""
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    tokens = torch.randn(B, L + 1, D, generator=g(1000 * rank + 10 * batch + 0))
    attn = torch.randn(B, L + 1, L + 1, generator=g(1000 * rank + 10 * batch + 3))
    return tokens.to(device), attn.to(device)


def make_inputs(rank, device):
    """(tokens, codebook, attn) of batch 0 - the batch of SURVEY.md 8(d); tools/ use it"""
    tokens, attn = make_batch(rank, device, 0)
    return tokens, make_codebook(device), attn


def make_model(device):
    import discretization
    import schema_inference.graph as graph
    torch.manual_seed(4)
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, dist_pow=2, feat_h=14, feat_w=14, clamp_vertex_attn=-1.0,
                         clamp_edge_attn=-1.0, remove_self_loop=False, prune_node_threshold=0.001)
    sn.register_class_vertices(torch.arange(M).repeat(K, 1))
    torch.manual_seed(5)
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    disc = discretization.Discretization(M, D)
    return disc.to(device), sn.to(device), m.to(device)


def FUSED_ATLAS(sn, width=E):
    """atlas normalisation feeding the GCN operand directly (no [K,n,n] class_edges round trip); a pruned atlas: the operand
    compacted to the kept vertices of each class (SchemaNet.get_atlas(fused_adjacency="compact"): as plain on an atlas
    without pruned vertices, e.g. the freshly initialised one of the main bench line)"""
    mode = False if os.environ.get("SN_FUSED_ATLAS", "1") == "0" else ("compact" if width == 256 else True)
    return lambda: sn.get_atlas(fused_adjacency=mode)


def step(disc, sn, m, tokens, attn, class_branch_first=True, side_stream=None, votes=None, defer=None):
    """class_branch_first: the class branch (parameters only) is forked before S1, so the replayed
    graph can fill S1's tail and the gaps of the instance chain with it (483 vs 509 us per step);
    the instrumented pass forks it behind S1 so that the S1 kernels are timed alone on the GPU.
    side_stream: see Matcher.atlas_features_async (False = the class branch in line on the step's own stream)."""
    # S1 = the fp16-MFMA screen + the fp64 finish of the tokens it cannot decide.  One step at a time (`defer`, as
    # SchemaNetPredictor.forward does): the finish rides in the instance-graph kernel's row phase - no re-rank launch on the
    # step's critical path (+2 % one at a time).  Several steps in flight: the stand-alone re-rank - a light kernel the
    # other steps' kernels run beside, where the fused form lengthens a kernel that holds every CU's LDS (-3.5 %).
    # SN_S1_DEFER=0 / 1 forces either.
    if defer is None:
        defer = side_stream is not False
    env = os.environ.get("SN_S1_DEFER", "")
    if env in ("0", "1"):
        defer = env == "1"
    if class_branch_first and os.environ.get("SN_CLASS_BRANCH_FIRST", "1") != "0":
        atlas = m.atlas_features_async(FUSED_ATLAS(sn, m.gnn.embed_dim), side_stream=side_stream)  # atlas normalise + class-graph GNN
        ing, rerank = disc.assign(tokens[:, 1:, :], defer=True) if defer else (disc.assign(tokens[:, 1:, :]), None)      # S1
    else:
        ing, rerank = disc.assign(tokens[:, 1:, :], defer=True) if defer else (disc.assign(tokens[:, 1:, :]), None)
        atlas = m.atlas_features_async(FUSED_ATLAS(sn, m.gnn.embed_dim), side_stream=side_stream)
    g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False,
                                 zero_padding=os.environ.get("SN_ZERO_PADDING", "0") == "1", rerank=rerank)   # S2 + S3 (as SchemaNetPredictor.forward)
    return m.forward_padded(g, atlas.class_dict, feat_kg=atlas, votes=votes)     # S4 (instance GNN, join, scores [+ the per-class votes])


def stream_copy_GBps(device, n_bytes=1 << 30, reps=5):
    """On-box HBM ceiling (report both denominators): device-to-device copy of 1 GiB, read + written
    bytes per second (MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy against the 8 TB/s spec)."""
    a = torch.empty(n_bytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n_bytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def event_pair_floor_ms(n=200):
    """What a HIP event pair reads with NOTHING between the two records on the launch stream (the two dispatch gaps
    every `avg_launch_ms` below contains; rocprofv3's kernel trace does not): median of n pairs."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        b.record()
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ev)
    return v[n // 2]


def kernel_times(lib, kid):
    n = lib.sn_profile_count(kid)
    if n == 0:
        return []
    buf = (ctypes.c_float * n)()
    rc = lib.sn_profile_elapsed_ms(kid, buf, n)
    return list(buf) if rc == 0 else []


def cpu_baseline(tokens, codebook, attn, sn, m, n_img=B):
    """The reference's CPU path on the host cores for a bounded sample (one batch of n_img images
    of the same workload, full atlas).  oracle/ is used here as the thing being timed as the
    BASELINE, never as the product."""
    from oracle import cpu_pipeline
    # torch intra-op threads capped at the PHYSICAL core count (round 6: 128 torch threads on a 256-CPU box made the cdist of the
    # discretize stage 3.5 x slower than 8 threads on 8 vCPUs, and the figure halved between rounds); the C++ graph stage is
    # single-threaded by construction, like the reference's (no OpenMP, GIL held: SURVEY 8b)
    nt_before = torch.get_num_threads()
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = max(1, (os.cpu_count() or 2) // 2)
    try:
        phys = min(phys, len(os.sched_getaffinity(0)))            # (a box may hand this process fewer CPUs than it has)
    except Exception:
        pass
    nt = max(1, min(nt_before, phys, 64))
    torch.set_num_threads(nt)
    P = {"gnn." + k: v.detach().cpu() for k, v in m.gnn.state_dict().items()}
    args = (tokens[:n_img].cpu(), attn[:n_img].cpu(), codebook.cpu(), sn.vertex_weights.tensor.detach().cpu(),
            sn.edge_weights.tensor.detach().cpu(), sn.class_ingredients.tensor.cpu(), P,
            sn.vertex_attribute_weights.tensor.detach().cpu(), sn.edge_attribute_weights.tensor.detach().cpu())
    try:
        for _ in range(2):
            cpu_pipeline.forward(*args)                   # warm-up (thread pools, page-in)
        times, t0 = [], time.perf_counter()
        stages = {}
        # the median of >= 10 warm iterations; bounded: stops early (never under 5) only when 45 s have gone by
        while len(times) < 10 or (time.perf_counter() - t0 < 12.0 and len(times) < 50):
            t1 = time.perf_counter()
            pred, ing, st = cpu_pipeline.forward(*args)
            times.append(time.perf_counter() - t1)
            for k_, v in st.items():
                stages.setdefault(k_, []).append(v)
            if len(times) >= 5 and time.perf_counter() - t0 > 45.0:
                break
    finally:
        torch.set_num_threads(nt_before)
    reps = len(times)
    dt = float(np.median(times))
    return {
        "value": n_img / dt, "unit": "images/sec", "cores": nt, "kind": cpu_pipeline.cpp_stage_kind(),
        "sample": f"median of {reps} timed repetitions (2 warm-ups before them) of one batch of {n_img} images of the same workload (full K=100, "
                  f"n_max=512 atlas per batch); torch ops on {nt} threads = min(torch default {nt_before}, physical cores {phys}, 64), "
                  f"C++ graph stage single-threaded like the reference",
        "host_cpus": os.cpu_count(), "physical_cores": phys, "repetitions": reps,
        "value_min_max": [n_img / max(times), n_img / min(times)],
        "stage_ms": {k_: 1e3 * float(np.median(v)) for k_, v in stages.items()},
    }, pred, ing


class _ResidentBackbone(torch.nn.Module):
    """Stands where the TorchScript backbone stands in the reference (`backbone_jit(x)` ->
    {"mid_feat": [L+1, bs, D] sequence-first, "extracted": [bs*H, L+1, L+1] raw attention logits},
    ingredient_model_wrapper.py:45-47): hands out batches that are already resident in HBM, in rotation (the backbone's
    own time is excluded from the metric)."""

    def __init__(self, batches):
        super().__init__()
        self.batches, self.i = batches, 0

    def forward(self, x):
        mid_feat, extracted = self.batches[self.i % len(self.batches)]
        self.i += 1
        return {"mid_feat": mid_feat, "extracted": extracted}


def api_leg(device, disc, sn, m, n_calls, n_batches=4):
    """img/s through `SchemaNetPredictor.forward` itself (reference schema_inference/graph/__init__.py:37-57; callers
    eval/evaluation.py:71, worker_schema_net.py:128-136): one call per batch, one at a time, raw backbone taps in."""
    import discretization
    import schema_inference.graph as graph
    from schema_inference.utils import IngredientModelWrapper
    g = lambda s_: torch.Generator().manual_seed(s_)  # noqa: E731
    batches = []
    for i in range(n_batches):
        mid = torch.randn(L + 1, B, D, generator=g(100 + 10 * i)).to(device)                 # sequence-first, cls row first
        ext = torch.randn(B * H, L + 1, L + 1, generator=g(103 + 10 * i)).to(device)         # randn(256*6,197,197)
        batches.append((mid, ext))
    # the same taps with the heads already averaged (one "head": what the main bench line feeds), to separate what the API
    # costs from what the per-head input costs (931 KB instead of 155 KB of attention per image)
    batches_h1 = [(mid, ext.reshape(B, H, L + 1, L + 1).mean(dim=1).contiguous()) for mid, ext in batches]
    backbone = _ResidentBackbone(batches)
    wrapper = IngredientModelWrapper(backbone, discretization.DiscretizationModule(disc))
    pred = graph.SchemaNetPredictor(wrapper, sn, m).eval()
    x = torch.empty(B, 3, 1, 1, device=device)                                               # (the images: unused by the stand-in)
    out = {}
    with torch.no_grad():
        for name, cache, bt, ring in (("value_api", False, batches, False), ("value_api_eval_cache", True, batches, False),
                                      ("value_api_head_averaged", False, batches_h1, False), ("value_api_head_averaged_ring", False, batches_h1, True)):
            backbone.batches = bt
            pred.output_ring = ring          # (opt-in: `pred` from a two-deep ring of captures instead of a copy - valid until the next call but one)
            pred.matcher.cache_atlas = cache
            pred.matcher.invalidate_atlas_cache()
            pred.invalidate_graphs()
            wrapper.backbone_jit.i = 0
            for _ in range(2 * n_batches):                                                   # captures + first replays
                last = pred(x)["pred"]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_calls):
                last = pred(x)["pred"]
            torch.cuda.synchronize()
            out[name] = B * n_calls / (time.perf_counter() - t1)
            out[name + "_replayed"] = bool(pred.graph_replay and len(pred._graphs) > 0)
        pred.output_ring = False
        assert tuple(last.shape) == (B, K) and bool(torch.isfinite(last).all())
        # the evaluation-loop form of the API: `predict_batches` keeps `depth` batches in flight by itself
        for name, cache, bt in (("value_api_batches", False, batches), ("value_api_batches_eval_cache", True, batches),
                                ("value_api_batches_head_averaged", False, batches_h1)):
            backbone.batches = bt
            pred.matcher.cache_atlas = cache
            pred.matcher.invalidate_atlas_cache()
            pred.invalidate_graphs()
            wrapper.backbone_jit.i = 0
            for o in pred.predict_batches((x for _ in range(2 * n_batches)), depth=n_batches):
                last = o["pred"]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for o in pred.predict_batches((x for _ in range(n_calls)), depth=n_batches):
                last = o["pred"]
            t_host = time.perf_counter() - t1
            torch.cuda.synchronize()
            out[name] = B * n_calls / (time.perf_counter() - t1)
            out[name + "_host_us_per_batch"] = 1e6 * t_host / n_calls       # time the submitting thread needs (no synchronisation in the loop)
        assert tuple(last.shape) == (B, K) and bool(torch.isfinite(last).all())
        pred.invalidate_graphs()
    out["api_note"] = (f"SchemaNetPredictor.forward(x) once per batch, one call at a time ({n_calls} calls over {n_batches} resident batches): "
                       "raw taps in (tokens [197,256,384] sequence-first, per-head logits [256*6,197,197]: the head mean is fused into the "
                       "instance-graph kernel, 931 KB of attention per image instead of 155 KB), the reference's dict out (pred + class_vertices "
                       "+ class_edges + class_ingredients); value_api: class-graph branch recomputed in every call (Matcher.cache_atlas off, "
                       "like value_depth1), value_api_eval_cache: the predictor's eval() default (class-graph features kept per parameter version), "
                       "value_api_head_averaged: as value_api with the head mean taken beforehand ([256,197,197] logits in, the input of `value` / "
                       "`value_depth1`): the API route itself against the hand-written step; the per-head input alone adds 199 MB of mandatory reads per call; "
                       f"value_api_batches(_head_averaged): `SchemaNetPredictor.predict_batches(loader, depth={n_batches})` - the same calls made by the "
                       "predictor's own evaluation loop, which keeps that many batches in flight on its own streams (what `value` does by hand)")
    return out


def shape_leg(device, name, Bc, Dc, Mc, Kc, n_max, Ec, token_dtype, n_steps):
    """The same step (S1 -> instance graph -> atlas branch -> matcher) at another configuration's shape, one step at a
    time from a capture with the class branch forked (eager launches when the configuration cannot be captured)."""
    import discretization
    import schema_inference.graph as graph
    from schema_inference.utils.graph_replay import GraphedStep
    g = lambda s_: torch.Generator().manual_seed(s_)  # noqa: E731
    pool = torch.randn(4 * Mc, Dc, generator=g(1))
    codebook = (pool[torch.randperm(4 * Mc, generator=g(11))[:Mc]] + 0.05 * torch.randn(Mc, Dc, generator=g(2))).to(device)
    torch.manual_seed(4)
    sn_c = graph.SchemaNet(num_vertices=Mc, num_classes=Kc, dist_pow=2, feat_h=14, feat_w=14, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0,
                           remove_self_loop=False, prune_node_threshold=0.001, class_max_vertices=n_max)
    sn_c.register_class_vertices(torch.stack([torch.randperm(Mc, generator=g(20 + k))[:n_max] for k in range(Kc)]))
    torch.manual_seed(5)
    m_c = graph.Matcher("inner_product", Mc, dict(embed_dim=Ec, num_layers=2, identity_proj=False, activation="relu"))
    disc_c = discretization.Discretization(Mc, Dc)
    disc_c, sn_c, m_c = disc_c.to(device), sn_c.to(device), m_c.to(device)
    with torch.no_grad():
        disc_c.vocabulary.weight.copy_(codebook)
    bt = [(torch.randn(Bc, L + 1, Dc, generator=g(200 + 10 * i)).to(device, token_dtype),
           torch.randn(Bc, L + 1, L + 1, generator=g(203 + 10 * i)).to(device)) for i in range(2)]
    fns = [(lambda tk=tk, at=at: step(disc_c, sn_c, m_c, tk, at, side_stream=True)) for tk, at in bt]
    how = "hipgraph, one step at a time, class branch forked"
    with torch.no_grad():
        try:
            caps = [GraphedStep(f) for f in fns]
            run = [c.replay for c in caps]
        except Exception as exc:                             # noqa: BLE001
            print(f"bench: {name}: capture failed ({exc!r}); eager launches", file=sys.stderr)
            run, how = fns, "eager launches"
        for i in range(4):
            last = run[i % 2]()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_steps):
            last = run[i % 2]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        assert tuple(last.shape) == (Bc, Kc) and bool(torch.isfinite(last).all())
        # per-kernel view (HIP event pairs inside the library, three eager steps in line): the S1 screen and the GCN products
        # against the dense f16 matrix peak - this is the configuration BASELINE attaches MFMA to
        lib = __import__("cpp_extension").load()
        n_prof = 3
        eager = lambda: step(disc_c, sn_c, m_c, bt[0][0], bt[0][1], class_branch_first=False, side_stream=False)      # noqa: E731
        eager()
        torch.cuda.synchronize()
        lib.sn_profile_enable(64 * n_prof)
        for _ in range(n_prof):
            eager()
        torch.cuda.synchronize()
        k_ms = {k_: kernel_times(lib, i) for k_, i in (("assign_screen", 0), ("assign_rerank", 1), ("instance_graph", 2), ("gcn_gemm", 4))}
        lib.sn_profile_enable(0)
    PEAK = 2.5e15                                                   # MI355X_MICROARCH.md: dense f16 / bf16 MFMA
    s1_flops = 2.0 * Bc * L * Dc * Mc
    n_cls = n_max
    gcn_flops = 2.0 * (2 * Kc * n_cls * n_cls * Ec + Kc * n_cls * Ec * Ec)      # class side, the reference's three products per graph (fp32 GEMMs there)
    s1_ms = sum(k_ms["assign_screen"]) / max(1, len(k_ms["assign_screen"]))
    gemm_ms = sum(k_ms["gcn_gemm"]) / n_prof                        # all six launches of a step (class side: 3, > 90 % of it)
    roof = {"bound": "mfma", "peak": PEAK / 1e12, "unit": "TFLOP/s",
            "s1_screen_ms": s1_ms, "s1_flops": s1_flops, "s1_TFLOPs": s1_flops / (s1_ms * 1e-3) / 1e12 if s1_ms else None,
            "s1_frac": s1_flops / (s1_ms * 1e-3) / PEAK if s1_ms else None,
            "s1_bytes": Bc * L * (Dc * (2 if token_dtype == torch.bfloat16 else 4) + 8),
            "gcn_gemm_ms_per_step": gemm_ms, "gcn_launches_per_step": len(k_ms["gcn_gemm"]) / n_prof,
            "gcn_class_flops_reference_formulation": gcn_flops,
            "gcn_useful_TFLOPs": gcn_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None,
            "gcn_frac_useful": gcn_flops / (gemm_ms * 1e-3) / PEAK if gemm_ms else None,
            "gcn_frac_issued": 3.0 * gcn_flops / (gemm_ms * 1e-3) / PEAK if gemm_ms else None,
            "note": "eager steps in line, HIP event pair per launch; gcn_frac_useful = the reference's fp32-GEMM flops of the class side over the time of "
                    "ALL GCN product launches of a step / 2.5 PF; gcn_frac_issued = x 3 (split-f16: hi.hi + hi.lo + lo.hi MFMAs per product, what keeps "
                    "the scores within 1e-5); rocprofv3 kernel tables of the same steps: profiles/r06_c4_kernel_stats.csv, r06_c5_kernel_stats.csv"}
    return {"value": Bc * n_steps / dt, "ms_per_step": 1e3 * dt / n_steps, "steps": n_steps, "launch": how, "roofline": roof,
            "shape": {"batch": Bc, "D": Dc, "words": Mc, "classes": Kc, "vertices_per_class": n_max, "gnn_width": Ec,
                      "tokens": str(token_dtype).replace("torch.", "")}}


def train_leg(device, n_iter=10):
    """One optimisation step of the reference's SchemaNet trainer at config [4]'s real size (B = 64, M = 1024, K = 101 class graphs
    of 1024 vertices - a 404 MB edge_weights -, E = 256; schema_inference/tasks/worker_schema_net.py:121-147): normalize -> forward ->
    SchemaInferenceLoss -> backward -> AdamW as ONE hipGraph replay (`train.GraphedTrainIter`), and the same iteration as eager
    `train_iter` calls over the same route.  Milliseconds per iteration; untimed by the driver."""
    import schema_inference.graph as graph
    from schema_inference import loss as loss_mod, train as train_mod
    Bc, Mc, Kc, Ec = 64, 1024, 101, 256
    g = lambda s_: torch.Generator().manual_seed(s_)  # noqa: E731
    ing = torch.randint(0, Mc, (Bc, L), generator=g(1))
    ing[:, ::3] = ing[:, :1]
    batch = {"ingredients": ing.to(device), "attn": torch.randn(Bc, L, L, generator=g(2)).to(device), "attn_cls": torch.randn(Bc, L, generator=g(3)).to(device)}
    target = {"label": torch.randint(0, Kc, (Bc,), generator=g(4)).to(device)}
    weights = {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}
    loss_fn = loss_mod.get_loss_fn({"name": "schema_inference_loss"})
    out = {}
    for how in ("graphed", "eager"):
        torch.manual_seed(11)
        sn_t = graph.SchemaNet(num_vertices=Mc, num_classes=Kc, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(device)
        sn_t.register_class_vertices(torch.arange(Mc, device=device).repeat(Kc, 1))
        torch.manual_seed(12)
        m_t = graph.Matcher("inner_product", Mc, dict(embed_dim=Ec, num_layers=2, identity_proj=False, activation="relu")).to(device).train()

        def fwd(b):                                              # the route SchemaNetPredictor takes behind its wrapper
            atlas = sn_t.get_atlas()
            o = {"pred": m_t.forward_padded(sn_t.instance_graph_padded(b["ingredients"], b["attn"].clone(), b["attn_cls"].clone()), atlas)}
            o.update(atlas)
            return o
        params = list(sn_t.parameters()) + list(m_t.parameters())
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-4, fused=True, capturable=how == "graphed")
        if how == "graphed":
            step_fn = train_mod.GraphedTrainIter(fwd, sn_t, loss_fn, weights, opt, batch, target, warmup=2)
            run = lambda: step_fn(batch, target)                 # noqa: E731
        else:
            run = lambda: train_mod.train_iter(lambda: fwd(batch), sn_t, loss_fn, weights, opt, target)   # noqa: E731
            for _ in range(2):
                run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_iter):
            total, _ = run()
        torch.cuda.synchronize()
        out["c5_train_iter_ms" + ("" if how == "graphed" else "_eager")] = 1e3 * (time.perf_counter() - t1) / n_iter
        assert bool(torch.isfinite(total))
        del sn_t, m_t, opt, run
        torch.cuda.empty_cache()
    # ---- the reference's OWN call sequence (worker_schema_net.py:121-147: optimizer.zero_grad(); schema_net.normalize(); output =
    # self.predictor(x); loss; backward; step), eager, through `SchemaNetPredictor.forward` in train() over a wrapper whose backbone
    # hands out resident taps (tokens that discretize to the same words, the same attention logits, one head): S1 included
    import discretization
    from schema_inference.utils import IngredientModelWrapper
    torch.manual_seed(11)
    sn_t = graph.SchemaNet(num_vertices=Mc, num_classes=Kc, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(device)
    sn_t.register_class_vertices(torch.arange(Mc, device=device).repeat(Kc, 1))
    torch.manual_seed(12)
    m_t = graph.Matcher("inner_product", Mc, dict(embed_dim=Ec, num_layers=2, identity_proj=False, activation="relu")).to(device)
    disc_t = discretization.Discretization(Mc, D).to(device)
    with torch.no_grad():
        disc_t.vocabulary.weight.copy_(torch.randn(Mc, D, generator=g(5)).to(device))
        cbk = disc_t.vocabulary.weight
        mid = torch.zeros(L + 1, Bc, D, device=device)
        mid[1:] = cbk[batch["ingredients"].t()] + 0.01 * torch.randn(L, Bc, D, generator=g(6)).to(device)
        ext = torch.zeros(Bc, L + 1, L + 1, device=device)
        ext[:, 1:, 1:] = batch["attn"]
        ext[:, 0, 1:] = batch["attn_cls"]
    wrapper = IngredientModelWrapper(_ResidentBackbone([(mid, ext)]), discretization.DiscretizationModule(disc_t))
    predictor = graph.SchemaNetPredictor(wrapper, sn_t, m_t).train()
    with torch.no_grad():
        assert torch.equal(wrapper.taps(torch.empty(Bc, 3, 1, 1, device=device))["ingredients"], batch["ingredients"])
    params = list(sn_t.parameters()) + list(m_t.parameters())
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-4, fused=True)
    x_img = torch.empty(Bc, 3, 1, 1, device=device)
    run = lambda: train_mod.train_iter(lambda: predictor(x_img), sn_t, loss_fn, weights, opt, target)      # noqa: E731
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_iter):
        total, _ = run()
    torch.cuda.synchronize()
    out["c5_train_iter_ms_reference_calls"] = 1e3 * (time.perf_counter() - t1) / n_iter
    assert bool(torch.isfinite(total))
    del sn_t, m_t, opt, run, predictor, wrapper
    torch.cuda.empty_cache()
    out["c5_train_note"] = ("config [4] at its real size, one training iteration (normalize, forward, loss, backward, fused AdamW): `train.GraphedTrainIter` "
                            "(one hipGraph launch; the class GNN on the kept vertices of the atlas, half of which is pruned at this initialisation) / eager `train_iter` "
                            "calls over the same padded batch (class graphs at full size, the instance pass on a second stream: the better route "
                            "when every launch costs host time); r03: 13.4 ms, r04: 6.5 ms; c5_train_iter_ms_reference_calls (round 6): the reference "
                            "trainer's own sequence - optimizer.zero_grad(), schema_net.normalize(), SchemaNetPredictor.forward(x) in train(), loss, "
                            "backward, step - eager, S1 on resident tokens included (worker_schema_net.py:121-147)")
    return out


def pruned_atlas_leg(device, disc, m, batches, depth, n_steps):
    """The step of `value` with a PRUNED IR-Atlas: a trained atlas is sparse - the loss's entropy terms push most vertices of a
    class under prune_node_threshold (reference schema_net.py:152-166) - while the freshly initialised one of the main line
    prunes nothing.  Same shapes (K = 100 classes x 512 vertices), ~70 % of every class's vertex weights under the threshold;
    the class branch is recomputed in every step as in `value`.  `value_pruned_atlas`: the GCN operand compacted to the kept
    vertices of each class (SchemaNet.get_atlas(fused_adjacency="compact")); `value_pruned_atlas_uncompacted`: the same
    atlas through the plain route (SN_ATLAS_COMPACT=0) - what `value` would be on it."""
    import schema_inference.graph as graph
    from schema_inference.utils.graph_replay import PipelinedSteps
    torch.manual_seed(4)
    sn_p = graph.SchemaNet(num_vertices=M, num_classes=K, dist_pow=2, feat_h=14, feat_w=14, clamp_vertex_attn=-1.0,
                           clamp_edge_attn=-1.0, remove_self_loop=False, prune_node_threshold=0.001)
    sn_p.register_class_vertices(torch.arange(M).repeat(K, 1))
    g = torch.Generator().manual_seed(44)
    with torch.no_grad():
        low = torch.rand(K, M, generator=g) < 0.7
        sn_p.vertex_weights.tensor.copy_(torch.where(low, sn_p.vertex_weights.tensor * 1.0e-4, sn_p.vertex_weights.tensor))
    sn_p = sn_p.to(device)
    out = {}
    old = os.environ.get("SN_ATLAS_COMPACT")
    try:
        with torch.no_grad():
            for name, flag in (("value_pruned_atlas", "1"), ("value_pruned_atlas_uncompacted", "0")):
                os.environ["SN_ATLAS_COMPACT"] = flag
                fns = [(lambda tk=tk, at=at: step(disc, sn_p, m, tk, at, side_stream=False)) for tk, at in batches]
                pipe = PipelinedSteps(fns, depth)
                for _ in range(2 * len(fns)):
                    last = pipe.submit()
                pipe.join()
                best = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(n_steps):
                        last = pipe.submit()
                    pipe.join()
                    torch.cuda.synchronize()
                    best.append(B * n_steps / (time.perf_counter() - t1))
                out[name] = sorted(best)[1]
                assert tuple(last.shape) == (B, K) and bool(torch.isfinite(last).all())
                del pipe
    finally:
        if old is None:
            os.environ.pop("SN_ATLAS_COMPACT", None)
        else:
            os.environ["SN_ATLAS_COMPACT"] = old
    with torch.no_grad():
        cv = sn_p.get_class_vertices(detach=True)
        out["pruned_atlas_note"] = (f"{float((cv <= 0.001).float().mean()):.2f} of the class vertices under prune_node_threshold; "
                                    f"{depth} steps in flight over {len(batches)} batches, median of three regions of {n_steps} steps, class branch "
                                    "recomputed in every step; compacted: per-class extents in the producer and both products, the isolated "
                                    "vertices' share of the class feature from the per-word table of GNN.prepare()")
    return out


def init_atlas_leg(device, rank, world, n_img=64):
    """One IR-Atlas initialisation (reference scripts/init_schema_net.py:105-124) over a synthetic shard of n_img
    images per rank, on a SchemaNet of its own: pass 1 (class vertex sums) -> merge -> top vertices -> pass 2 (class edge
    sums [K, n_max, n_max] = 105 MB) -> merge -> normalise.  With world > 1 the two merges are the RCCL collectives of
    the per-class schema statistics; HIP events bracket them."""
    import schema_inference.graph as graph
    from schema_inference.graph.statistics import SchemaStatistics
    torch.manual_seed(40)
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, dist_pow=2, feat_h=14, feat_w=14, clamp_vertex_attn=-1.0,
                         clamp_edge_attn=-1.0, remove_self_loop=False, prune_node_threshold=0.001).to(device)
    g = torch.Generator().manual_seed(7000 + rank)
    ing = torch.randint(0, M, (n_img, L), generator=g).to(device)
    attn = torch.randn(n_img, L, L, generator=g).to(device)
    acls = torch.randn(n_img, L, generator=g).to(device)
    label = ((torch.arange(n_img) * world + rank) % K).to(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    with torch.no_grad():
        for timed in (False, True):                          # first pass: warm-up (kernels, communicator buffers)
            stats = SchemaStatistics(K, M, M, device=device)
            torch.cuda.synchronize()
            ev[0].record()
            stats.add_vertices(sn.feat_to_full_vertices(ing, acls.clone()), label)
            ev[1].record()
            stats.all_reduce_vertices()
            ev[2].record()
            init_w, valid = stats.top_vertices()
            sn.register_class_vertices(valid)
            sn.vertex_weights.copy_(init_w)
            stats.add_edges(sn.feat_to_limited_edges(ing, attn, label), label)
            ev[3].record()
            stats.all_reduce_edges()
            ev[4].record()
            sn.edge_weights.copy_(stats.class_edges())
            sn.normalize()
            ev[5].record()
            torch.cuda.synchronize()
    n_e = stats._edges().numel()
    e_ms = ev[3].elapsed_time(ev[4])
    return {
        "images_per_rank": n_img, "world_size": world, "total_ms": ev[0].elapsed_time(ev[5]),
        "vertex_stats_collective_ms": ev[1].elapsed_time(ev[2]), "edge_stats_collective_ms": e_ms,
        "edge_stats_collective": stats.last_collective, "edge_stats_bytes": n_e * 4,
        # bus bandwidth of an all-reduce: 2 (W-1)/W x bytes / time
        "edge_stats_busbw_GBps": (2.0 * (world - 1) / world * n_e * 4 / (e_ms * 1e-3) / 1e9) if world > 1 and e_ms > 0 else None,
        "note": "outside the timed region; reference scripts/init_schema_net.py:19-65 (two passes, one merge each)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--regions", type=int, default=7, help="timed regions of --steps steps each; value = the median region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip value_api / c1_value / c4_value (N = 1 only anyway)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # SN_BENCH_REHEARSAL=1: every rank on device 0 over gloo - the N > 1 code path of this file (barriers, vote merge,
    # init_atlas merges, max over ranks) on a one-GPU box, where RCCL cannot form a group; its numbers mean nothing
    rehearsal = os.environ.get("SN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its version banner on stdout when the communicator is created: keep stdout for the JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=device)       # "nccl" is RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist_world = dist.get_world_size() if use_dist else 1

    import cpp_extension
    from cpp_extension import ops
    lib = cpp_extension.load()
    n_batches = max(1, int(os.environ.get("SN_BENCH_BATCHES", "8")))
    depth = max(1, int(os.environ.get("SN_BENCH_DEPTH", "4")))
    if n_batches % depth != 0:
        n_batches = depth * ((n_batches + depth - 1) // depth)
    codebook = make_codebook(device)
    batches = [make_batch(rank, device, i) for i in range(n_batches)]     # distinct (tokens, attn) per capture
    tokens, attn = batches[0]
    disc, sn, m = make_model(device)
    with torch.no_grad():
        disc.vocabulary.weight.copy_(codebook)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    votes = torch.zeros(K + 1, device=device)            # per-class prediction histogram + n_seen

    def step_on(i, side_stream):
        tk, at = batches[i]

        def one_step():
            # (per-class vote aggregation in the launch that writes the scores: HIP, no host sync; SN_BENCH_VOTES_FUSED=0: its own launch)
            if os.environ.get("SN_BENCH_VOTES_FUSED", "1") == "0":
                pred = step(disc, sn, m, tk, at, side_stream=side_stream)
                ops.class_votes_(pred, votes)
                return pred
            return step(disc, sn, m, tk, at, side_stream=side_stream, votes=votes)
        return one_step

    # Several steps in flight: each step runs in line on its own stream (the device maps streams onto four hardware
    # queues; four independent serial steps use them without cross-queue waits: 764 k img/s against 736 k with the class
    # branch of every step forked onto a stream of its own).  One step at a time: the fork is what fills the chip
    # (600 k against 524 k).  SN_BENCH_FORK=0/1 forces either for the pipelined captures.
    fork_env = os.environ.get("SN_BENCH_FORK", "")
    pipe_fork = (fork_env == "1") if fork_env in ("0", "1") else depth == 1
    steps_fn = [step_on(i, pipe_fork) for i in range(n_batches)]
    launch = "eager"
    value_depth1 = value_depth1_inline = None
    with torch.no_grad():
        for w in range(args.warmup):
            steps_fn[w % n_batches]()
        graphed = None
        if os.environ.get("SN_BENCH_EAGER", "0") != "1":
            try:
                from schema_inference.utils.graph_replay import GraphedStep, PipelinedSteps
                graphed = PipelinedSteps(steps_fn, depth)   # one capture per batch (outside the timed region)
                # captures of the same steps with the class branch forked: what ONE stream of batches should run
                forked = graphed.steps if pipe_fork else [GraphedStep(step_on(i, True)) for i in range(n_batches)]
                for i in range(n_batches):                  # every captured graph replayed once (first replay = upload)
                    graphed.steps[i].graph.replay()
                    if forked is not graphed.steps:
                        forked[i].graph.replay()

                n_d1 = max(args.steps, 100)                 # (a leg of 20 replays is 7 ms: too short to time well)

                def one_at_a_time(caps):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for s in range(n_d1):
                        caps[s % n_batches].graph.replay()
                    torch.cuda.synchronize()
                    return B * n_d1 / (time.perf_counter() - t1)
                # ---- the same K steps, one in flight at a time (depth 1): what a single stream of batches gets.
                # These legs run BEFORE the pipelined one: the captures above are host work with an idle GPU, and
                # the timed region of a short run (the driver's --steps 20) would otherwise start on idle clocks
                if depth > 1:
                    value_depth1 = one_at_a_time(forked)
                    if forked is not graphed.steps:
                        value_depth1_inline = one_at_a_time(graphed.steps)
                for _ in range(args.warmup):                # W untimed steps of the timed kind (pipelined replays)
                    graphed.submit()
                graphed.join()
                launch = f"hipgraph, {n_batches} batches in rotation, {depth} steps in flight" if depth > 1 else f"hipgraph, {n_batches} batches in rotation"
                launch += ", class branch forked onto a second stream" if pipe_fork else ", class branch in line"
            except Exception as exc:                     # noqa: BLE001 - fall back to eager launches, and say so
                print(f"bench: hipGraph capture failed ({exc!r}); timing eager launches", file=sys.stderr)
                graphed = None
        region_dt, n_voted = [], 0
        for _ in range(max(1, args.regions)):
            votes.zero_()
            barrier()
            t0 = time.perf_counter()
            for s in range(args.steps):
                if graphed is not None:
                    graphed.submit()
                else:
                    steps_fn[s % n_batches]()
            if graphed is not None:
                graphed.join()
            if use_dist:
                dist.all_reduce(votes)                       # eval-meter merge over RCCL
            barrier()
            region_dt.append(time.perf_counter() - t0)
            n_voted = int(votes[K].item())
            assert n_voted == B * args.steps * world, (n_voted, B * args.steps * world)

        # ---- untimed instrumented pass: the same K steps launched eagerly, HIP events around the
        # kernels (inside the library, on the launch stream) and around the stages
        lib.sn_profile_enable(args.steps)
        stage_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        ings = {}
        for s in range(args.steps):
            tk, at = batches[s % n_batches]
            ev = stage_ev[s]
            ev[0].record()
            ing = disc.assign(tk[:, 1:, :])                  # S1 runs alone (the side stream starts behind it)
            ev[1].record()
            atlas = m.atlas_features_async(FUSED_ATLAS(sn))  # class branch on the side stream (overlaps everything below)
            ev[2].record()
            g = sn.instance_graph_padded(ing, at[:, 1:, 1:], at[:, 0, 1:], mutate_inputs=False, zero_padding=False)
            ev[3].record()
            pred = m.forward_padded(g, atlas.class_dict, feat_kg=atlas)
            ev[4].record()
            ings[s % n_batches] = ing
        torch.cuda.synchronize()
        k_ms = {name: kernel_times(lib, kid) for kid, name in enumerate(("assign_screen", "assign_rerank", "instance_graph", "atlas_normalize", "gcn_gemm"))}
        # S2+S3 alone on the GPU, like S1 above (in the pass it runs beside the class branch of the side stream, and how much
        # of that branch it meets depends on how fast the host enqueued it: 36 us on one day, 56 us on another)
        lib.sn_profile_enable(args.steps)
        for s in range(args.steps):
            tk, at = batches[s % n_batches]
            sn.instance_graph_padded(ings[s % n_batches], at[:, 1:, 1:], at[:, 0, 1:], mutate_inputs=False, zero_padding=False)
        torch.cuda.synchronize()
        k_ms["instance_graph_beside_class_branch"] = k_ms["instance_graph"]
        k_ms["instance_graph"] = kernel_times(lib, 2)
        atlas_leg = init_atlas_leg(device, rank, dist_world)
        extra = {}
        if world == 1 and not args.no_extra_legs:
            extra.update(api_leg(device, disc, sn, m, n_calls=max(args.steps, 100)))
            c1 = shape_leg(device, "c1", 32, 192, 128, 10, 128, 256, torch.float32, max(args.steps, 100))
            c4 = shape_leg(device, "c4", 256, 768, 1024, 1000, 500, 1024, torch.bfloat16, 10)
            c5 = shape_leg(device, "c5", 64, 384, 1024, 101, 1024, 256, torch.float32, 20)
            extra.update(pruned_atlas_leg(device, disc, m, batches, depth, max(args.steps, 40)))
            with torch.enable_grad():
                extra.update(train_leg(device))
            extra.update({"c1_value": c1["value"], "c1": c1, "c4_value": c4["value"], "c4": c4, "c5_value": c5["value"], "c5": c5,
                          "roofline_c4": c4["roofline"], "roofline_c5": c5["roofline"]})
    t_max = torch.tensor(region_dt, device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)         # per region: the slowest rank
    region_dt = sorted(t_max.tolist())
    dt = region_dt[len(region_dt) // 2]                      # the median region

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        stage_ms = [sum(stage_ev[s][i].elapsed_time(stage_ev[s][i + 1]) for s in range(args.steps)) / args.steps
                    for i in range(4)]
        avg = {k_: (sum(v) / len(v) if v else None) for k_, v in k_ms.items()}
        # roofline of the assignment kernel (north_star): algorithmic bytes per launch =
        # B*196 tokens * (D*4 B read + 8 B index written)  (302,624 B / image)
        alg_bytes = B * L * (D * 4 + 8)
        ach = alg_bytes / (avg["assign_screen"] * 1e-3) / 1e9 if avg["assign_screen"] else None
        s1_ms = (avg["assign_screen"] or 0.0) + (avg["assign_rerank"] or 0.0)
        # S2+S3: attn + ids + cls attention in; the images' own n_i x n_i edge corners (the zero padding is not written) + padded ids / weights out
        graph_bytes = B * (L * L * 4 + L * 4 + L * 8) + int((g["n"].long() ** 2).sum().item()) * 4 + B * L * 12
        g_ach = (graph_bytes / (avg["instance_graph"] * 1e-3) / 1e9) if avg["instance_graph"] else None
        variant = lib.sn_assign_variant()
        screen_name = {5: "assign_screen5_kernel<12,0> (S1 fp16-MFMA screen, K-outer one-round form: opt-in, SN_ASSIGN_VARIANT=5)"}.get(
            variant, "assign_screen_kernel<24,4,3,8,false> (S1 fp16-MFMA screen, token-stationary)")
        copy_gbps = stream_copy_GBps(device)
        ev_floor = event_pair_floor_ms()
        traffic, traffic_s3, traffic_src = None, None, None      # HBM bytes per launch from the committed rocprofv3 PMC passes (not measurable live)
        for name in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    for row in json.load(fh)["kernels"]:
                        if "::" + screen_name.split(" ")[0].split("<")[0] + "<" in row["kernel"] or row["kernel"].startswith(screen_name.split("<")[0] + "<"):
                            traffic = row["hbm_read_bytes_corrected"] + row["hbm_write_bytes"]
                        if "instance_graph_kernel<true" in row["kernel"] or "instance_graph_kernel<1" in row["kernel"]:        # (<true, true>: the prediction configuration)
                            traffic_s3 = row["hbm_read_bytes_corrected"] + row["hbm_write_bytes"]
                if traffic is not None:
                    traffic_src = f"profiles/{name} (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
                    break
            except (OSError, KeyError, ValueError):
                pass
        # whole step: bytes that must cross HBM once (tokens, attention, the atlas parameters, the ids / scores out) and the
        # dense flops of the reference's formulation (S1 distance matrix + the six GCN products as fp32 GEMMs)
        n_mean = float(g["n"].float().mean().item())
        step_bytes = B * (L + 1) * D * 4 + B * (L + 1) * (L + 1) * 4 + K * M * M * 4 + K * M * 4 + B * L * 8 + B * K * 4
        gcn_flops = 2 * (2 * K * M * M * E + K * M * E * E) + 2 * (2 * B * L * L * E + B * L * E * E)
        s1_flops = 2 * B * L * D * M
        out = {
            "metric": "images/sec schema-inference (discretize+graph) DeiT-S CIFAR-100",
            "value": B * world * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "regions": len(region_dt), "value_min": B * world * args.steps / region_dt[-1], "value_max": B * world * args.steps / region_dt[0],
            "region_note": f"{len(region_dt)} back-to-back timed regions of exactly {args.steps} steps (barrier + synchronize on both sides of each, max over ranks per region): value / ms_per_step = the median region", "launch": launch + (" [REHEARSAL: all ranks on one GPU over gloo - not a measurement]" if rehearsal else ""),
            "value_depth1": value_depth1, "value_depth1_inline": value_depth1_inline,
            "value_note": f"value: {depth} steps in flight on {depth} streams over {n_batches} distinct resident batches ({n_batches} x 117 MB of inputs per GPU), "
                          f"every step {'with its class branch forked onto a second stream' if pipe_fork else 'in line on its own stream'}; "
                          "value_depth1: the same steps one at a time (max(K, 100) replays), captured with the class branch forked onto a second stream (what one "
                          "stream of batches should run); value_depth1_inline: the captures of `value` replayed one at a time (rank 0 only, untimed by the driver)",
            "vs_baseline": None, "dtype": "f32 (S1 screen: f16 MFMA + f64 re-rank; GCN: split-f16 MFMA, f32 accumulate; ids int64)", "data": "synthetic",
            "world_size": dist_world, "votes_merged": n_voted,
            "config": {"workload": "configs[1]: DeiT-Small + CIFAR-100, synthetic [256,197,384] tokens per GPU, "
                                   "512-word codebook, head-averaged attention logits [256,197,197], K=100, "
                                   "n_max=512, GNN E=256 x 2 layers; atlas recomputed every step",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"image-parallel x{world}",
                       "resident_batches": n_batches, "steps_in_flight": depth, "feat_kg_cache": "off",
                       "caches_across_steps": ["packed_codebook (fp16 fragment image of the 512 words, per codebook version)",
                                               "gnn_prepared (Emb . W1^T table, W2 planes, fc^T, the per-word feature of an isolated vertex: weight-only "
                                               "operands, per weight version; the reference runs linear(embedding) in every forward)",
                                               "atlas_facts (per version of the IR-Atlas parameters: whether any class vertex is under the prune "
                                               "threshold - then the class branch runs compacted -, and that the in-place pruning has run on these "
                                               "versions - then the rows it zeroed are not read again; the values of the atlas are re-read every step)"],
                       "s1_finish": "one step at a time (value_depth1, value_api*): inside the instance-graph kernel; several steps in flight "
                                    "(value, value_api_batches*): the stand-alone re-rank kernel - the faster form in either regime (DESIGN 3.1e)"},
            "roofline": {"bound": "hbm", "kernel": screen_name,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg["assign_screen"],
                         "whole_assignment_ms": s1_ms,
                         "whole_assignment_frac": (alg_bytes / (s1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if s1_ms else None,
                         "event_pair_floor_ms": ev_floor,
                         "avg_launch_note": "avg_launch_ms = HIP event pair around the kernel on its launch stream (what achieved / frac use); an event pair with nothing between reads event_pair_floor_ms, so the kernel trace of rocprofv3 (profiles/) shows this kernel ~that much shorter; whole_assignment = screen + fp64 re-rank",
                         "peak_note": "peak = 8.0 TB/s HBM3E spec; copy_GBps = a 1 GiB device-to-device copy on this box (read + write)",
                         "copy_GBps": copy_gbps, "frac_of_copy": (ach / copy_gbps) if (ach and copy_gbps) else None,
                         "matrix_flops_per_launch": s1_flops,
                         "matrix_TFLOPs": (s1_flops / (avg["assign_screen"] * 1e-3) / 1e12) if avg.get("assign_screen") else None,
                         "matrix_frac_of_2.5PF": (s1_flops / (avg["assign_screen"] * 1e-3) / 2.5e15) if avg.get("assign_screen") else None,
                         "matrix_note": "the screen multiplies every token with every word: 2 x tokens x words x D flops of f16 MFMA per launch take "
                                        "7.9 us at the dense peak (~11 us at the clock the pipe holds under load) against 9.7 us for its bytes at 8 TB/s: on this "
                                        "workload (no bound prunes a word) the matrix pipe is a second roofline at the height of the HBM one (DESIGN 3.1d)"},
            "roofline_s3": {"bound": "hbm", "kernel": "instance_graph_kernel<true, true> (S2+S3: grouping, edge cells, normalise; compile-time prediction configuration)",
                            "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (g_ach / HBM_PEAK_GBS) if g_ach else None,
                            "traffic": traffic_s3, "algorithmic_bytes_per_launch": graph_bytes, "avg_launch_ms": avg["instance_graph"],
                            "avg_launch_note": "launched alone (HIP event pair inside the library); beside the class branch of the side stream, "
                                               "as in the instrumented pass: kernels_ms.instance_graph_beside_class_branch",
                            "mean_vertices_per_image": n_mean},
            "roofline_step": {"algorithmic_bytes_per_step": step_bytes, "hbm_GBps": step_bytes / (ms_step * 1e-3) / 1e9,
                              "hbm_frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "dense_flops_per_step": s1_flops + gcn_flops, "TFLOPs": (s1_flops + gcn_flops) / (ms_step * 1e-3) / 1e12,
                              "mfma_frac_of_2.5PF": (s1_flops + gcn_flops) / (ms_step * 1e-3) / 2.5e15,
                              "note": "bytes that must cross HBM once per step (tokens, attention logits, IR-Atlas parameters, ids and scores out) and the dense flops of the REFERENCE's formulation (token x codebook distances + its six GCN products: two adjacency products and one Linear per layer and side), over ms_per_step of the timed region; this build runs five of the six per side-pair (layer 1's Linear is folded into the embedding table once per weight version, see config.caches_across_steps) and the split-f16 products issue 3 MFMAs per dense flop pair"},
            "kernels_ms": avg,
            "instance_graph_GBps": g_ach,
            "stage_ms": dict(zip(("S1_assign", "atlas_branch_enqueue", "S2S3_instance_graph", "S4_instance_gnn_join_scores"), stage_ms)),
            "stage_note": "main-stream intervals of the instrumented eager pass (slower than the timed hipGraph replays: event records + host launches); the class branch (atlas normalise + GNN over K graphs) runs concurrently on a side stream and is joined inside S4",
            "init_atlas": atlas_leg,
        }
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            cb, pred_cpu, ing_cpu = cpu_baseline(tokens, codebook, attn, sn, m)
            out["cpu_baseline"] = cb
            # sanity: the CPU pipeline and the GPU path agree on the sample (the parity test proper is tests/test_gpu_parity.py)
            with torch.no_grad():
                n_img = pred_cpu.shape[0]
                ing_gpu = disc.assign(tokens[:n_img, 1:, :]).cpu()
                out["cpu_baseline"]["word_id_mismatches_vs_gpu"] = int((ing_gpu != ing_cpu).sum())
                pred_gpu = step(disc, sn, m, tokens, attn)[:n_img].cpu()
                scale = float(pred_cpu.abs().max())
                # (max |difference| over max |score|: a scale-relative figure; the element-wise criterion is tests/test_gpu_parity.py::scores_close)
                out["cpu_baseline"]["pred_max_abs_err_over_score_scale_vs_gpu"] = float((pred_gpu - pred_cpu).abs().max()) / scale
                out["cpu_baseline"]["top1_mismatches_vs_gpu"] = int((pred_gpu.argmax(1) != pred_cpu.argmax(1)).sum())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    lib.sn_profile_enable(0)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
