"""GPU (-m gpu), more than one card: the statistics collective on RCCL.  Skipped on a one-GPU box (the gloo tests
of test_host_cpu.py cover the same code path on the CPU backend)."""
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
