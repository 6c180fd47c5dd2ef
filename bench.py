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
Nothing is cached across steps (the atlas GCN is recomputed every step, like the reference).
The K timed steps replay captured hipGraphs of the step (schema_inference.utils.graph_replay: same kernels,
no host launch path in the timed region), four steps in flight on four streams (independent batches; each capture
has its own buffers; `SN_BENCH_DEPTH=1` replays one graph back to back); a
second, untimed pass of K eager steps with HIP events on the launch stream gives the per-kernel
durations of the roofline figure (`SN_BENCH_EAGER=1` times the eager loop instead).
Workload = BASELINE.json configs[1]: DeiT-Small + CIFAR-100, B=256 per GPU, 512-word codebook.
Multi-GPU: images are sharded over ranks (weak scaling, B per rank fixed), no data-path
collective; the per-class prediction histogram + (n_seen) are all-reduced once at the end of
the timed region over RCCL (the eval-meter merge of the reference, eval/evaluation.py:95-97).
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

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
B, L, D, M, K, E, H = 256, 196, 384, 512, 100, 256, 6


def make_inputs(rank, device):
    """seeded CPU generators, then copied (same bits on every box).  Every rank
    gets its own images (seed offset) but the same codebook / atlas / matcher."""
    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    tokens = torch.randn(B, L + 1, D, generator=g(1000 * rank + 0))
    pool = torch.randn(4096, D, generator=g(1))
    codebook = pool[torch.randperm(4096, generator=g(11))[:M]] + 0.05 * torch.randn(M, D, generator=g(2))
    attn = torch.randn(B, L + 1, L + 1, generator=g(1000 * rank + 3))
    return tokens.to(device), codebook.to(device), attn.to(device)


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


def FUSED_ATLAS(sn):
    """atlas normalisation feeding the GCN operand directly (no [K,n,n] class_edges round trip)"""
    return lambda: sn.get_atlas(fused_adjacency=os.environ.get("SN_FUSED_ATLAS", "1") != "0")


def step(disc, sn, m, tokens, attn, class_branch_first=True):
    """class_branch_first: the class branch (parameters only) is forked before S1, so the replayed
    graph can fill S1's tail and the gaps of the instance chain with it (483 vs 509 us per step);
    the instrumented pass forks it behind S1 so that the S1 kernels are timed alone on the GPU."""
    if class_branch_first and os.environ.get("SN_CLASS_BRANCH_FIRST", "1") != "0":
        atlas = m.atlas_features_async(FUSED_ATLAS(sn))                          # side stream: atlas normalise + class-graph GNN
        ing = disc.assign(tokens[:, 1:, :])                                      # S1
    else:
        ing = disc.assign(tokens[:, 1:, :])
        atlas = m.atlas_features_async(FUSED_ATLAS(sn))
    g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False,
                                 zero_padding=os.environ.get("SN_ZERO_PADDING", "0") == "1")   # S2 + S3 (as SchemaNetPredictor.forward)
    return m.forward_padded(g, atlas.class_dict, feat_kg=atlas)                  # S4 (instance GNN, join, scores)


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
    nt = torch.get_num_threads()
    P = {"gnn." + k: v.detach().cpu() for k, v in m.gnn.state_dict().items()}
    args = (tokens[:n_img].cpu(), attn[:n_img].cpu(), codebook.cpu(), sn.vertex_weights.tensor.detach().cpu(),
            sn.edge_weights.tensor.detach().cpu(), sn.class_ingredients.tensor.cpu(), P,
            sn.vertex_attribute_weights.tensor.detach().cpu(), sn.edge_attribute_weights.tensor.detach().cpu())
    cpu_pipeline.forward(*args)                       # warm-up (thread pools, page-in)
    reps, t0 = 0, time.perf_counter()
    stages = {}
    while reps < 2 or (time.perf_counter() - t0 < 12.0 and reps < 50):
        pred, ing, st = cpu_pipeline.forward(*args)
        for k_, v in st.items():
            stages[k_] = stages.get(k_, 0.0) + v
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {
        "value": n_img / dt, "unit": "images/sec", "cores": nt, "kind": cpu_pipeline.cpp_stage_kind(),
        "sample": f"{reps} x one batch of {n_img} images of the same workload (full K=100, n_max=512 atlas "
                  f"per batch); torch ops on {nt} threads, C++ graph stage single-threaded like the reference",
        "host_cpus": os.cpu_count(),
        "stage_ms": {k_: 1e3 * v / reps for k_, v in stages.items()},
    }, pred, ing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
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
            dist.init_process_group("nccl", device_id=device)           # "nccl" is RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import cpp_extension
    from cpp_extension import ops
    lib = cpp_extension.load()
    tokens, codebook, attn = make_inputs(rank, device)
    disc, sn, m = make_model(device)
    with torch.no_grad():
        disc.vocabulary.weight.copy_(codebook)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    votes = torch.zeros(K + 1, device=device)            # per-class prediction histogram + n_seen

    def one_step():
        pred = step(disc, sn, m, tokens, attn)
        ops.class_votes_(pred, votes)                    # per-class vote aggregation (HIP, no host sync)
        return pred

    launch = "eager"
    with torch.no_grad():
        for _ in range(args.warmup):
            one_step()
        graphed = None
        if os.environ.get("SN_BENCH_EAGER", "0") != "1":
            try:
                from schema_inference.utils.graph_replay import PipelinedSteps
                depth = int(os.environ.get("SN_BENCH_DEPTH", "4"))
                graphed = PipelinedSteps(one_step, depth)   # capture (outside the timed region)
                for _ in range(depth):
                    graphed.submit()
                graphed.join()
                launch = f"hipgraph, {depth} steps in flight" if depth > 1 else "hipgraph"
            except Exception as exc:                     # noqa: BLE001 - fall back to eager launches, and say so
                print(f"bench: hipGraph capture failed ({exc!r}); timing eager launches", file=sys.stderr)
                graphed = None
        votes.zero_()
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            if graphed is not None:
                graphed.submit()
            else:
                one_step()
        if graphed is not None:
            graphed.join()
        if use_dist:
            dist.all_reduce(votes)                       # per-class schema statistics over RCCL
        barrier()
        dt = time.perf_counter() - t0
        n_voted = int(votes[K].item())

        # ---- untimed instrumented pass: the same K steps launched eagerly, HIP events around the
        # kernels (inside the library, on the launch stream) and around the stages
        lib.sn_profile_enable(args.steps)
        stage_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        for s in range(args.steps):
            ev = stage_ev[s]
            ev[0].record()
            ing = disc.assign(tokens[:, 1:, :])              # S1 runs alone (the side stream starts behind it)
            ev[1].record()
            atlas = m.atlas_features_async(FUSED_ATLAS(sn))  # class branch on the side stream (overlaps everything below)
            ev[2].record()
            g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False, zero_padding=False)
            ev[3].record()
            pred = m.forward_padded(g, atlas.class_dict, feat_kg=atlas)
            ev[4].record()
        torch.cuda.synchronize()
    t_max = torch.tensor([dt], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    assert n_voted == B * args.steps * world, (n_voted, B * args.steps * world)

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        stage_ms = [sum(stage_ev[s][i].elapsed_time(stage_ev[s][i + 1]) for s in range(args.steps)) / args.steps
                    for i in range(4)]
        k_ms = {name: kernel_times(lib, kid) for kid, name in enumerate(("assign_screen", "assign_rerank", "instance_graph", "atlas_normalize"))}
        avg = {k_: (sum(v) / len(v) if v else None) for k_, v in k_ms.items()}
        # roofline of the assignment kernel (north_star): algorithmic bytes per launch =
        # B*196 tokens * (D*4 B read + 8 B index written)  (302,624 B / image)
        alg_bytes = B * L * (D * 4 + 8)
        ach = alg_bytes / (avg["assign_screen"] * 1e-3) / 1e9 if avg["assign_screen"] else None
        # attn + ids + cls attention in; the images' own n_i x n_i edge corners (the zero padding is not written) + padded ids / weights out
        graph_bytes = B * (L * L * 4 + L * 4 + L * 8) + int((g["n"].long() ** 2).sum().item()) * 4 + B * L * 12
        screen_name = ("assign_screen2_kernel<4,24> (S1 fp16-MFMA screen, codebook-stationary)" if lib.sn_assign_variant() == 2
                       else "assign_screen_kernel<24,4,3> (S1 fp16-MFMA screen, token-stationary)")
        copy_gbps = stream_copy_GBps(device)
        ev_floor = event_pair_floor_ms()
        traffic = None          # HBM bytes per launch from the committed rocprofv3 PMC passes (not measurable live)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as fh:
                for row in json.load(fh)["kernels"]:
                    if row["kernel"].startswith(screen_name.split(" ")[0].split(",")[0]):
                        traffic = row["hbm_read_bytes_corrected"] + row["hbm_write_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "images/sec schema-inference (discretize+graph) DeiT-S CIFAR-100",
            "value": B * world * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "launch": launch,
            "vs_baseline": None, "dtype": "f32 (S1 screen: f16 MFMA + f64 re-rank; ids int64)", "data": "synthetic",
            "config": {"workload": "configs[1]: DeiT-Small + CIFAR-100, synthetic [256,197,384] tokens per GPU, "
                                   "512-word codebook, head-averaged attention logits [256,197,197], K=100, "
                                   "n_max=512, GNN E=256 x 2 layers; atlas recomputed every step",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"image-parallel x{world}"},
            "roofline": {"bound": "hbm", "kernel": screen_name,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_hbm_traffic.json (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg["assign_screen"],
                         "event_pair_floor_ms": ev_floor,
                         "avg_launch_note": "avg_launch_ms = HIP event pair around the kernel on its launch stream (what achieved / frac use); an event pair with nothing between reads event_pair_floor_ms, so the kernel trace of rocprofv3 (profiles/) shows this kernel ~that much shorter",
                         "peak_note": "peak = 8.0 TB/s HBM3E spec; copy_GBps = a 1 GiB device-to-device copy on this box (read + write)",
                         "copy_GBps": copy_gbps, "frac_of_copy": (ach / copy_gbps) if (ach and copy_gbps) else None},
            "kernels_ms": avg,
            "instance_graph_GBps": (graph_bytes / (avg["instance_graph"] * 1e-3) / 1e9) if avg["instance_graph"] else None,
            "stage_ms": dict(zip(("S1_assign", "atlas_branch_enqueue", "S2S3_instance_graph", "S4_instance_gnn_join_scores"), stage_ms)),
            "stage_note": "main-stream intervals of the instrumented eager pass (slower than the timed hipGraph replays: event records + host launches); the class branch (atlas normalise + GNN over K graphs) runs concurrently on a side stream and is joined inside S4",
        }
        if not args.no_cpu_baseline and world == 1:
            cb, pred_cpu, ing_cpu = cpu_baseline(tokens, codebook, attn, sn, m)
            out["cpu_baseline"] = cb
            # sanity: the CPU pipeline and the GPU path agree on the sample (not a parity test)
            with torch.no_grad():
                n_img = pred_cpu.shape[0]
                ing_gpu = disc.assign(tokens[:n_img, 1:, :]).cpu()
                out["cpu_baseline"]["word_id_mismatches_vs_gpu"] = int((ing_gpu != ing_cpu).sum())
                pred_gpu = step(disc, sn, m, tokens, attn)[:n_img].cpu()
                scale = float(pred_cpu.abs().max())
                out["cpu_baseline"]["pred_max_rel_err_vs_gpu"] = float((pred_gpu - pred_cpu).abs().max()) / scale
                out["cpu_baseline"]["top1_mismatches_vs_gpu"] = int((pred_gpu.argmax(1) != pred_cpu.argmax(1)).sum())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    lib.sn_profile_enable(0)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
