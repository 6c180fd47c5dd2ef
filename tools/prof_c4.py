"""GPU diagnostic: kernel breakdown of one step at config [3]'s shape (D = 768 bf16 tokens, 1024 words, K = 1000 classes of 500
vertices, GNN width 1024, 256 images) - run under `rocprofv3 --kernel-trace --stats` (from /tmp), or alone for event times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
r = bench.shape_leg(dev, "c4", 256, 768, 1024, 1000, 500, 1024, torch.bfloat16, n)
print(r)
