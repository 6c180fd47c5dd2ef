"""GPU diagnostic: which ops of a training iteration issue hipMemsetAsync (a captured memset node is unreliable on this ROCm -
csrc/sn_common.h, sn_zero_async).  python tools/find_memsets.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "2", "1", "1", "padded"]
import torch
from torch.profiler import profile, ProfilerActivity
src = open(os.path.join(ROOT, "tools", "prof_c5_train.py")).read()
ns = {"__name__": "prof", "__file__": os.path.join(ROOT, "tools", "prof_c5_train.py")}
exec(compile(src, "prof_c5_train.py", "exec"), ns)
model, batch, target, loss_fn, weights, opt, train_mod = (ns[k] for k in ("model", "batch", "target", "loss_fn", "weights", "opt", "train_mod"))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_mod.train_iter(lambda: model(batch), model.schema_net, loss_fn, weights, opt, target)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if "emset" in ev.name:
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        cnt[(ev.name, " <- ".join(chain))] += 1
for k, v in cnt.most_common():
    print(v, k)
