"""GPU (-m gpu): the collectives on RCCL.  The two-rank test needs two cards (skipped on a one-GPU box; the gloo tests of
test_host_cpu.py and the two-rank rehearsal of test_gpu_api.py cover the same code on the CPU backend); the one-rank tests
run the SAME calls through RCCL itself - process-group set-up bound to the device, barrier, all_reduce, reduce_scatter_tensor +
all_gather_into_tensor of the statistics buffers - which is what a one-GPU box can show of the N > 1 path."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "schemanet-pytorch_amd"))
rank, world = int(sys.argv[3]), int(sys.argv[4])
force_single = len(sys.argv) > 5
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world, device_id=dev)
import schema_inference.graph as graph
from schema_inference.graph.statistics import init_atlas
B, L, M, K, n_max = 16, 196, 64, 5, 37            # K*n_max^2 + K = 6850: not a multiple of the world size
g = torch.Generator().manual_seed(0)
ing = torch.randint(0, M, (B, L), generator=g); attn = torch.randn(B, L, L, generator=g); acls = torch.randn(B, L, generator=g)
label = torch.arange(B) % K

def run(idx, all_reduce, large_bytes):
    torch.manual_seed(1)
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, class_max_vertices=n_max, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0,
                         prune_node_threshold=0.001).to(dev)
    graph.SchemaStatistics.large_bytes = large_bytes
    graph.SchemaStatistics.merge_single_rank = force_single
    stats = init_atlas(sn, [(ing[idx].to(dev), attn[idx].to(dev), acls[idx].to(dev), label[idx].to(dev))], all_reduce=all_reduce)
    return sn, stats

whole, _ = run(torch.arange(B), False, 8 << 20)
ok = True
for large_bytes, want in ((1 << 10, "reduce_scatter+all_gather"), (1 << 40, "all_reduce")):
    part, stats = run(graph.shard_indices(B, rank, world), True, large_bytes)
    ok &= stats.last_collective == want
    ok &= torch.equal(part.class_ingredients.tensor, whole.class_ingredients.tensor)
    ok &= torch.allclose(part.vertex_weights.tensor, whole.vertex_weights.tensor, rtol=1e-5, atol=1e-8)
    ok &= torch.allclose(part.edge_weights.tensor.nan_to_num(), whole.edge_weights.tensor.nan_to_num(), rtol=1e-5, atol=1e-8)
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_statistics_collective_two_ranks_rccl(tmp_path):
    """init_atlas on two image shards merged over RCCL == init_atlas on the whole set, on both forms of the merge."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(33500 + os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), "2"], env=env) for r in range(2)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0, 0], codes


def test_statistics_collectives_one_rank_rccl(tmp_path):
    """The worker above with a world of one: both forms of the merge go through RCCL (reduce_scatter_tensor + all_gather_into_tensor
    on the padded flat buffer, all_reduce) and leave the statistics of the single shard unchanged."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(35500 + os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT, port, "0", "1", "merge"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]


def test_bench_one_rank_over_rccl():
    """`bench.py` as the driver launches it for N > 1 (torch.distributed.run, RANK / MASTER_* in the environment), with one rank:
    the "nccl" (= RCCL) process group bound to the device, the barriers around the timed regions, the all_reduce of the vote
    vector and of the region times, the statistics merges of the IR-Atlas initialisation leg - one JSON line, world_size 1."""
    steps, port = 3, str(31600 + os.getpid() % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", SN_BENCH_BATCHES="4")
    env.pop("SN_BENCH_REHEARSAL", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "2",
           "--regions", "2", "--no-cpu-baseline", "--no-extra-legs"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["world_size"] == 1 and out["steps"] == steps
    assert out["votes_merged"] == 256 * steps
    assert out["init_atlas"]["world_size"] == 1
    assert out["value"] > 0 and "REHEARSAL" not in out["launch"]
