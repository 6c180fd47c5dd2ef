import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "schemanet-pytorch_amd"))
import torch, bench, cpp_extension
lib = cpp_extension.load()
dev = torch.device("cuda", 0)
tokens, codebook, attn = bench.make_inputs(0, dev)
disc, sn, m = bench.make_model(dev)
def run(n, torch_events, prof):
    lib.sn_profile_enable(n if prof else 0)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(n):
        if torch_events: evs[s][0].record()
        ing = disc.assign(tokens[:, 1:, :])
        if torch_events: evs[s][1].record()
        g = sn.instance_graph_padded(ing, attn[:, 1:, 1:], attn[:, 0, 1:], mutate_inputs=False)
        if torch_events: evs[s][2].record()
        atlas = sn.get_atlas()
        if torch_events: evs[s][3].record()
        pred = m.forward_padded(g, atlas)
        if torch_events: evs[s][4].record()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    lib.sn_profile_enable(0)
    return dt * 1e3
with torch.no_grad():
    disc.vocabulary.weight.copy_(codebook)
    run(5, False, False)
    for te, pf in ((False, False), (True, False), (False, True), (True, True), (False, False)):
        print("torch_events=%s sn_profile=%s: %.3f ms/step" % (te, pf, run(30, te, pf)))
    votes = torch.zeros(bench.K + 1, device=dev); ones = torch.ones(bench.B, device=dev); n_img = torch.full((1,), 256.0, device=dev)
    for name in ("index_add", "scatter_add", "argmax_only", "none"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(30):
            pred = bench.step(disc, sn, m, tokens, attn)
            if name == "index_add":
                votes.index_add_(0, pred.argmax(dim=1), ones); votes[bench.K:] += n_img
            elif name == "scatter_add":
                votes.scatter_add_(0, pred.argmax(dim=1), ones); votes[bench.K:] += n_img
            elif name == "argmax_only":
                a = pred.argmax(dim=1)
        torch.cuda.synchronize(); print(name, "%.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3))
