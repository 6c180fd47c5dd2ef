"""Round 6: what each stage of the bench step costs IN THE PIPELINE (four captured steps in flight, as `value` is measured): the step with
one stage at a time replaced by its precomputed result.  Per variant: img/s and the microseconds per step the stage costs there -
next to its kernel time alone, this says whether `value` follows the kernels' own durations or something they share (clock, HBM).
    python tools/marginal_stage_cost.py [steps per region]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import bench
from schema_inference.utils.graph_replay import PipelinedSteps

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = bench.B
codebook = bench.make_codebook(dev)
batches = [bench.make_batch(0, dev, i) for i in range(8)]
disc, sn, m = bench.make_model(dev)
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)


def variant(skip):
    pre = []
    with torch.no_grad():
        for tk, at in batches:
            ing = disc.assign(tk[:, 1:, :])
            g = sn.instance_graph_padded(ing, at[:, 1:, 1:], at[:, 0, 1:], mutate_inputs=False, zero_padding=False)
            atlas = m.atlas_features_async(bench.FUSED_ATLAS(sn, m.gnn.embed_dim), side_stream=False)
            atlas.join()
            pre.append((ing, g, atlas))
    torch.cuda.synchronize()

    def step_on(i):
        tk, at = batches[i]
        ing0, g0, atlas0 = pre[i]

        def one():
            atlas = atlas0 if "class" in skip else m.atlas_features_async(bench.FUSED_ATLAS(sn, m.gnn.embed_dim), side_stream=False)
            ing = ing0 if "s1" in skip else disc.assign(tk[:, 1:, :])
            g = g0 if "s3" in skip else sn.instance_graph_padded(ing, at[:, 1:, 1:], at[:, 0, 1:], mutate_inputs=False, zero_padding=False)
            if "inst" in skip:
                return atlas.join() if hasattr(atlas, "join") else atlas
            return m.forward_padded(g, atlas.class_dict, feat_kg=atlas)
        return one
    with torch.no_grad():
        fns = [step_on(i) for i in range(8)]
        for f in fns:
            f()
        pipe = PipelinedSteps(fns, 4)
        for s_ in pipe.steps:
            s_.graph.replay()
        for _ in range(16):
            pipe.submit()
        pipe.join()
        vals = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                pipe.submit()
            pipe.join()
            torch.cuda.synchronize()
            vals.append((time.perf_counter() - t0) / n_steps)
    vals.sort()
    return vals[len(vals) // 2]


base = variant(())
print(f"full step: {1e6 * base:7.1f} us per step = {B / base / 1e3:7.1f} k img/s  (library: {os.environ.get('SN_LIB_PATH', 'default')}, SN_GEMM_TM={os.environ.get('SN_GEMM_TM', 'auto')})", flush=True)
if os.environ.get("SN_MARGINAL_BASE_ONLY") == "1":
    sys.exit(0)
for skip in (("s1",), ("s3",), ("class",), ("inst",), ("s1", "s3"), ("s1", "s3", "inst"), ("class", "inst")):
    t = variant(skip)
    print(f"without {'+'.join(skip):12s}: {1e6 * t:7.1f} us per step = {B / t / 1e3:7.1f} k img/s; costs {1e6 * (base - t):6.1f} us per step in the pipeline", flush=True)
