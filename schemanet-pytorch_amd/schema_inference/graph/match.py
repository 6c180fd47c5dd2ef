"""Instance-graph vs class-graph matching (reference schema_inference/graph/match.py)."""
import os
from typing import Any, Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from cpp_extension import ops

from .gnn import GNN


class _AtlasHandle:
    """Class-graph features being computed on the side stream (Matcher.atlas_features_async)."""

    def __init__(self, class_dict, feat, done, prepared=None):
        self.class_dict, self.feat, self.done, self.prepared = class_dict, feat, done, prepared

    def join(self) -> torch.Tensor:
        if self.done is not None:
            main = torch.cuda.current_stream(self.feat.device)
            main.wait_event(self.done)
            for v in [self.feat] + list(self.class_dict.values()):
                for t in ((v.hi, v.lo) if hasattr(v, "hi") else (tuple(v) if isinstance(v, (tuple, list)) else (v,))):
                    if torch.is_tensor(t):
                        t.record_stream(main)                        # allocated on the side stream, consumed here
            self.done = None
        return self.feat


def _quiet_stream_mismatch_in_backward(out: torch.Tensor) -> torch.Tensor:
    """The GNN's parameters of a side-stream iteration receive gradients from two streams; the engine orders them, and its warning
    about the mismatch is about an unintended one.  The process-wide switch is turned off for THIS backward pass only (it
    used to stay off for the rest of the process): a hook on the scores - the first gradient of the pass - saves and clears it, a
    callback queued on the engine restores it when the pass is over."""
    setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    getter = getattr(torch._C, "_warn_on_accumulate_grad_stream_mismatch", None)
    if setter is None or getter is None or not out.requires_grad:
        return out

    def at_backward_start(grad):
        before = bool(getter())
        if before:
            setter(False)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: setter(True))
        return grad
    out.register_hook(at_backward_start)
    return out


class Matcher(nn.Module):
    """Same constructor and state-dict (`gnn.*`) as the reference.  `forward` keeps the
    reference's list-based contract; `forward_padded` consumes the padded batch produced by
    `SchemaNet.instance_graph_padded` without python loops or host syncs."""

    def __init__(self, similarity: str, num_codes: int, gnn_cfg: Dict[str, Any]):
        super().__init__()
        if similarity not in ops.SIMILARITY:
            raise KeyError(similarity)
        self.similarity_name = similarity
        self.gnn = GNN(num_codes=num_codes, **gnn_cfg)
        # Class-graph features cache (inference only): the reference recomputes the GNN over the K
        # class graphs in every forward (its match.py:66-70) although it depends on parameters only.  With
        # `cache_atlas = True` (SchemaNetPredictor.eval() turns it on, .train() off) `atlas_features_async` keeps the
        # last result and returns it while no parameter it was computed from has changed: keyed on the (data_ptr,
        # _version) of the IR-Atlas tensors the caller names and of every GNN parameter.  In-place writes that bypass
        # the version counter (`p.data.copy_`) need `invalidate_atlas_cache()`.
        self.cache_atlas = False
        self._atlas_cache = None

    def __getstate__(self):
        """copies and pickles of the module (`copy.deepcopy` for an EMA model, `torch.save(model)`) leave the HIP streams and the
        cached class-branch handle behind: they are per-process resources, re-created on first use"""
        state = self.__dict__.copy()
        for k in ("_side_stream", "_train_stream", "_atlas_cache"):
            if k in state:
                state[k] = None
        return state

    # reference match.py:21-31
    def similarity(self, feat_inst: torch.Tensor, feat_kg: torch.Tensor, votes: torch.Tensor = None) -> torch.Tensor:
        """feat_inst [bs, E], feat_kg [K, E] -> [bs, K].  votes (f32 [K + 1], optional): the evaluation loop's per-class vote
        counter (reference eval/evaluation.py:81-97 via its meter), updated with the argmax of every image in the same launch."""
        if feat_inst.is_cuda and not (torch.is_grad_enabled() and (feat_inst.requires_grad or feat_kg.requires_grad)):
            return ops.match_scores(feat_inst, feat_kg, self.similarity_name, votes=votes)
        a, b = feat_inst[:, None, :], feat_kg[None, :, :]
        if self.similarity_name == "inner_product":
            pred = (a * b).sum(-1)
        elif self.similarity_name == "cosine":
            pred = (torch.cosine_similarity(a, b, dim=-1) + 1) / 2
        else:
            pred = 1 / (1 + torch.linalg.vector_norm(a - b, dim=-1))
        if votes is not None:
            ops.class_votes_(pred.detach(), votes)
        return pred

    def atlas_features(self, class_dict: Dict[str, torch.Tensor], prepared=None) -> torch.Tensor:
        """GNN over the K class graphs -> [K, E]  (reference match.py:66-70)."""
        if "class_adjacency" in class_dict:          # fused atlas route (no class_edges tensor)
            compact = (class_dict["class_perm"], class_dict["class_n_kept"]) if "class_perm" in class_dict else None
            return self.gnn(nodes=class_dict["class_vertices"], edges=None, ingredients=class_dict["class_ingredients"],
                            adjacency=class_dict["class_adjacency"], prepared=prepared, compact=compact)
        # (training with a pruned atlas: SchemaNet.get_atlas adds the partition of every class into kept and pruned vertices, and the
        # GNN runs its products on the kept ones - GNN.forward, "compacted class graphs")
        compact = (class_dict["class_perm"], class_dict["class_n_kept"]) if "class_perm" in class_dict and torch.is_grad_enabled() else None
        return self.gnn(nodes=class_dict["class_vertices"], edges=class_dict["class_edges"],
                        ingredients=class_dict["class_ingredients"], prepared=prepared, compact=compact)

    # ---- atlas branch on its own HIP stream -------------------------------------------------
    # The class-graph branch (atlas normalisation -> GNN over K graphs) depends only on parameters,
    # the instance branch (S1 -> instance graphs -> GNN over the batch) only on the images; they
    # meet at the similarity.  Without autograd the class branch runs on a side stream so that the
    # two chains fill each other's gaps (most kernels of either chain leave HBM or the matrix pipe
    # half idle); the result is joined with an event, no host synchronisation.
    def invalidate_atlas_cache(self):
        self._atlas_cache = None

    def _atlas_key(self, depends_on):
        """tensors by (data_ptr, _version, device); anything else in `depends_on` (scalar options of the caller, e.g.
        prune threshold / self-loop flag) by value"""
        ts = list(depends_on) + list(self.gnn.parameters())
        return tuple((t.data_ptr(), t._version, t.device) if torch.is_tensor(t) else ("opt", t) for t in ts)

    def atlas_features_async(self, get_class_dict, depends_on=None, side_stream=None):
        """`side_stream`: True = the class branch is forked onto a second HIP stream and overlaps the instance chain of
        the same forward pass (shortest latency of ONE pass: 600 k vs 524 k img/s at C2, one pass at a time); False = it
        runs in line on the current stream - the better choice when several passes are in flight on streams of their
        own (764 k vs 736 k img/s with four: the device maps streams onto four hardware queues, and four independent
        serial passes use them without cross-queue waits); None = True unless SN_SIDE_STREAM=0.

        `depends_on`: the tensors `get_class_dict()` reads (e.g. schema_net's vertex_weights / edge_weights /
        class_ingredients) and the scalar options it applies (any hashable non-tensor entries): with `cache_atlas` on and
        no autograd, the previous handle is returned while they and the GNN parameters are unchanged (one GNN pass over
        the K class graphs per parameter version instead of per forward).  Never cached while gradients could flow: grad
        mode on and a GNN parameter OR one of the `depends_on` tensors requiring grad (a cached handle would carry an
        autograd graph into later forwards).

        Start `get_class_dict()` (e.g. `schema_net.get_atlas`) + the class-graph GNN on the side
        stream.  Returns a handle for `forward_padded(..., feat_kg=handle)`; `handle.class_dict` is
        usable on the current stream after the join.  The weight-only operands of the GNN
        (`GNN.prepare`) are computed once, on the current stream before the fork, and shared by the
        class branch and the instance branch of this forward pass (`handle.prepared`)."""
        dev = next(self.gnn.parameters()).device
        grad_srcs = list(self.gnn.parameters()) + [t for t in (depends_on or ()) if torch.is_tensor(t)]
        no_grad = not (torch.is_grad_enabled() and any(p.requires_grad for p in grad_srcs))
        use_cache = self.cache_atlas and depends_on is not None and no_grad and dev.type == "cuda"
        if use_cache:
            key = self._atlas_key(depends_on)
            if self._atlas_cache is not None and self._atlas_cache[0] == key:
                return self._atlas_cache[1]
            handle = self._atlas_features_async(get_class_dict, dev, side_stream)
            handle.join()                                   # ordered behind the side stream once; later forwards just read it
            self._atlas_cache = (key, handle)
            return handle
        return self._atlas_features_async(get_class_dict, dev, side_stream)

    def _atlas_features_async(self, get_class_dict, dev, side_stream=None):
        prepared = self.gnn.prepare() if dev.type == "cuda" else None
        serial = (os.environ.get("SN_SIDE_STREAM", "1") == "0") if side_stream is None else not side_stream
        if serial or dev.type != "cuda" or torch.is_grad_enabled() and any(p.requires_grad for p in self.gnn.parameters()):
            class_dict = get_class_dict()
            return _AtlasHandle(class_dict, self.atlas_features(class_dict, prepared), None, prepared)
        if getattr(self, "_side_stream", None) is None or self._side_stream.device != dev:
            # (SN_CLASS_STREAM_PRIORITY: -1 = high; the class branch is the longer of the two chains of a step taken one at a time)
            self._side_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("SN_CLASS_STREAM_PRIORITY", "0")))
        main = torch.cuda.current_stream(dev)
        self._side_stream.wait_stream(main)                 # parameters / earlier work are visible
        for v in (prepared or {}).values():                # allocated on the current stream, also read on the side stream
            for t in ((v.hi, v.lo) if hasattr(v, "hi") else (tuple(v) if isinstance(v, (tuple, list)) else (v,))):
                if torch.is_tensor(t):
                    t.record_stream(self._side_stream)
        with torch.cuda.stream(self._side_stream):
            class_dict = get_class_dict()
            feat = self.atlas_features(class_dict, prepared)
            done = torch.cuda.Event()
            done.record(self._side_stream)
        return _AtlasHandle(class_dict, feat, done, prepared)

    def forward_padded(self, graph: Dict[str, torch.Tensor], class_dict: Dict[str, torch.Tensor],
                       feat_kg=None, votes: torch.Tensor = None) -> torch.Tensor:
        """graph: ids [bs, n_pad], vertices [bs, n_pad], edges [bs, n_pad, n_pad], n [bs] i32,
        n_max [1] i32 (device).  The pooling divides by n_max, i.e. by the length the reference
        pads to (max_i n_i, match.py:46; gnn.py:96), so the result does not depend on n_pad.
        feat_kg: precomputed class features [K, E] or a handle of `atlas_features_async`."""
        edges = graph["edges"]
        if not graph.get("edges_padded", True) and not self.gnn.masks_adjacency(edges):
            # written without its zero padding, and this GNN route reads all of it: mask by the vertex counts here
            valid = torch.arange(edges.shape[-1], device=edges.device)[None, :] < graph["n"][:, None]
            edges = torch.where(valid[:, :, None] & valid[:, None, :], edges, torch.zeros((), dtype=edges.dtype, device=edges.device))
        training = bool(feat_kg is None and edges.is_cuda and torch.is_grad_enabled() and any(p.requires_grad for p in self.gnn.parameters()))
        shared = {"train_table": self.gnn.train_table()} if training and self.gnn.train_folds() else None
        run_instance = lambda: self.gnn(nodes=graph["vertices"], edges=edges, ingredients=graph["ids"],     # noqa: E731
                                        n_valid=graph["n"], divisor=graph["n_max"],
                                        prepared=feat_kg.prepared if isinstance(feat_kg, _AtlasHandle) else shared)
        # (not beside a COMPACTED class pass: that one is a chain of small launches as well, and the two chains were measured slower
        # side by side - 5.55 ms - than one after the other - 5.23 ms - at config [4]'s real size)
        side_env = os.environ.get("SN_TRAIN_SIDE_STREAM", "1")     # 0: never, 1: beside an uncompacted class pass, 2: always
        # (gradients of the GNN's parameters then arrive from two streams.  The plain engine orders them; a hook that launches
        # work from the hook's current stream - a gradient all-reduce - has not been exercised with it, so parameters that carry hooks
        # keep both passes on one stream; DistributedDataParallel hooks the accumulation nodes, which cannot be seen from here:
        # INTEGRATION.md says to set SN_TRAIN_SIDE_STREAM=0 under it)
        if training and side_env == "1" and any(getattr(p_, "_backward_hooks", None) or getattr(p_, "_post_accumulate_grad_hooks", None)
                                                for p_ in self.gnn.parameters()):
            side_env = "0"
        if training and side_env != "0" and (side_env == "2" or "class_perm" not in class_dict):
            # Training: the two GNN passes of an iteration meet at the similarity only.  The instance pass is a chain of ~25 small
            # launches forward and ~50 backward (64 graphs of <= 196 vertices: ~20 us each whatever their size), the class pass a chain
            # of large ones: the instance pass runs on a second stream, forward AND backward (autograd runs a node's backward on the
            # stream of its forward and orders the streams where gradients cross), so its launches fill the gaps of the class pass -
            # eagerly and inside a captured iteration (train.GraphedTrainIter) alike.
            dev = edges.device
            if getattr(self, "_train_stream", None) is None or self._train_stream.device != dev:
                self._train_stream = torch.cuda.Stream(device=dev)
            main, side = torch.cuda.current_stream(dev), self._train_stream
            side.wait_stream(main)
            for t in (graph["vertices"], edges, graph["ids"], graph["n"], graph["n_max"], shared["train_table"] if shared else None):
                if torch.is_tensor(t):
                    t.record_stream(side)                    # allocated on the current stream, read on the side stream
            with torch.cuda.stream(side):
                feat_instance = run_instance()
            feat_kg = self.atlas_features(class_dict, shared)
            main.wait_stream(side)
            feat_instance.record_stream(main)
            return _quiet_stream_mismatch_in_backward(self.similarity(feat_instance, feat_kg, votes))
        feat_instance = run_instance()
        if isinstance(feat_kg, _AtlasHandle):
            feat_kg = feat_kg.join()
        elif feat_kg is None:
            feat_kg = self.atlas_features(class_dict, shared)
        return self.similarity(feat_instance, feat_kg, votes)

    @staticmethod
    def _padded_batch_of(instance_dict, ids, vs, es, sizes, n):
        """The batch padded to its maximum, without a launch, when the lists are the untouched views `SchemaNet.forward` made
        of its padded kernel output (schema_net.InstanceLists.padded: pad id / zeros in the padding); the callers' lists
        are padded IN PLACE as the reference does - with views of that batch.  None: the general route."""
        p = getattr(instance_dict, "padded", None)
        if p is None or p["sizes"] != sizes or len(ids) != p["ids"].shape[0]:
            return None
        for lst, base in ((ids, p["ids"]), (vs, p["vertices"]), (es, p["edges"])):
            for b, t in enumerate(lst):
                if t._base is not base or t.storage_offset() != base.storage_offset() + b * base.stride(0) or t.stride() != base.stride()[1:]:
                    return None
        graph = {"ids": p["ids"][:, :n], "vertices": p["vertices"][:, :n], "edges": p["edges"][:, :n, :n], "n": p["n"],
                 "n_max": torch.tensor([n], dtype=torch.int32, device=p["ids"].device)}
        for b in range(len(sizes)):
            ids[b], vs[b], es[b] = graph["ids"][b], graph["vertices"][b], graph["edges"][b]
        return graph

    def forward(self, instance_dict: Dict[str, List[torch.Tensor]], class_dict: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Reference contract (match.py:33-76): ragged python lists in, [bs, K] out.  Like the
        reference it pads the caller's lists IN PLACE to the batch maximum."""
        ids, vs, es = (instance_dict["instance_ingredients"], instance_dict["instance_vertices"],
                       instance_dict["instance_edges"])
        sizes = [len(x) for x in ids]
        n = max(sizes)
        graph = self._padded_batch_of(instance_dict, ids, vs, es, sizes, n)
        if graph is not None:
            return self.forward_padded(graph, class_dict)
        for i, s in enumerate(sizes):
            ids[i] = F.pad(ids[i], (0, n - s), value=self.gnn.num_codes)
            vs[i] = F.pad(vs[i], (0, n - s))
            es[i] = F.pad(es[i], (0, n - s, 0, n - s))
        dev = ids[0].device
        n_valid = torch.tensor(sizes, dtype=torch.int32, device=dev)
        graph = {"ids": torch.stack(ids), "vertices": torch.stack(vs), "edges": torch.stack(es),
                 "n": n_valid, "n_max": torch.tensor([n], dtype=torch.int32, device=dev)}
        return self.forward_padded(graph, class_dict)
