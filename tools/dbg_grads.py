"""GPU diagnostic: element-wise relative error of the 15 parameter gradients of the train-step fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "schemanet-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import datagen
import schema_inference.graph as graph
from schema_inference.loss import get_loss_fn
from schema_inference.train import weighted_total
DEV = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "train_step.npz")))
B, L, M, seed, K, n_max, E = g["case"].tolist()
ing, attn, attn_cls = datagen.graph_case(B, L, M, seed)
for mfma in ("1", "0"):
    os.environ["SN_GCN_MFMA"] = mfma
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, class_max_vertices=n_max, clamp_vertex_attn=-1.0, clamp_edge_attn=-1.0, prune_node_threshold=0.001).to(DEV)
    sn.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sn:")})
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu")).to(DEV)
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("m:")})
    sn.train(); m.train(); sn.normalize()
    with torch.enable_grad():
        pg = sn.instance_graph_padded(T(ing), T(attn), T(attn_cls))
        atlas = sn.get_atlas()
        pred = m.forward_padded(pg, atlas)
        ld = get_loss_fn({"name": "schema_inference_loss"})({"pred": pred, "class_vertices": atlas["class_vertices"], "class_edges": atlas["class_edges"]}, {"label": T(g["label"])})
        weighted_total(ld, {"cls": 1.0, "re_entropy_vertex": 0.5, "re_entropy_edge": 0.75}).backward()
    params = dict(list(sn.named_parameters()) + [("matcher." + n, q) for n, q in m.named_parameters()])
    print("SN_GCN_MFMA =", mfma)
    for k_, want in g.items():
        if not k_.startswith("grad:"):
            continue
        got = params[k_[5:]].grad.detach().cpu().numpy().astype(np.float64)
        w = want.astype(np.float64)
        sc = np.abs(w).max()
        big = np.abs(w) > 1e-3 * sc
        rel = np.abs(got - w)[big] / np.abs(w)[big] if big.any() else np.zeros(1)
        zero_ref = (w == 0)
        print("  %-46s scale %.2e  max|err|/scale %.1e  rel err (|g|>1e-3 scale): p50 %.1e p99 %.1e max %.1e | ref==0: %d, ours nonzero there: %d (max %.1e)"
              % (k_[5:], sc, np.abs(got - w).max() / sc, np.median(rel), np.quantile(rel, 0.99), rel.max(), zero_ref.sum(), (got[zero_ref] != 0).sum(),
                 np.abs(got[zero_ref]).max() if zero_ref.any() else 0))
