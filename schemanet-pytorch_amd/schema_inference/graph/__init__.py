"""`schema_inference.graph` -- the reference's public names (graph/__init__.py:7-9, 14-57),
MI355X-native.  `to_networkx` (visualisation helper) is outside the hot path and not provided.
"""
import collections
import os
from typing import Dict

import torch
import torch.nn as nn

from .schema_net import SchemaNet
from .match import Matcher
from .gnn import GNN
from .statistics import SchemaStatistics, shard_indices

__all__ = ["SchemaNet", "Matcher", "GNN", "SchemaNetPredictor", "SchemaStatistics", "shard_indices"]


class LazyOutputs(collections.OrderedDict):
    """The reference's output dictionary with entries that are only computed when somebody reads them.  `class_edges`
    [K, n, n] is 105 MB at config [1] (419 MB at 1024 words): the reference returns it from every forward, its evaluation
    loop (eval/evaluation.py:63-80) never looks at it - it is what the sparsity terms of the TRAINING loss read.  In eval()
    under no_grad the predictor therefore hands out the key with the value deferred: reading it (`out["class_edges"]`, `.get`,
    `.items()`, `.values()`) runs the atlas normalisation then, from the parameters as they are at that moment (the same
    values unless they were written in between).  Keys, their order, `in`, `len` and iteration over keys never compute."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}

    def set_lazy(self, key, fn):
        super().__setitem__(key, None)
        self._lazy[key] = fn

    def _force(self, key=None):
        for k in ([key] if key is not None else list(self._lazy)):
            fn = self._lazy.pop(k, None)
            if fn is not None:
                super().__setitem__(k, fn())

    def __getitem__(self, key):
        self._force(key)
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        if getattr(self, "_lazy", None):
            self._lazy.pop(key, None)
        super().__setitem__(key, value)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        if key in self:
            self._force(key)
        return super().pop(key, *default)

    def items(self):
        self._force()
        return super().items()

    def values(self):
        self._force()
        return super().values()

    def lazy_copy(self):
        """a copy that keeps the deferred entries deferred (`OrderedDict(self)` would compute them)"""
        out = LazyOutputs()
        for k in self.keys():
            if k in self._lazy:
                out.set_lazy(k, self._lazy[k])
            else:
                collections.OrderedDict.__setitem__(out, k, collections.OrderedDict.__getitem__(self, k))
        return out


class SchemaNetPredictor(nn.Module):
    """images -> ingredients -> instance IR-graph -> match against the IR-Atlas.

    forward returns the reference's OrderedDict: `pred` [bs, K], `class_vertices`,
    `class_edges`, `class_ingredients` and, with requires_graph, `instance_ingredients`,
    `instance_vertices`, `instance_edges` (lists, padded to the batch maximum exactly as the
    reference returns them after Matcher's in-place padding), `ingredients`, `attn_cls`.

    When the wrapper offers `taps(x)` (ours does) the whole path after the backbone is fused:
    word assignment straight from the sequence-first tokens, head mean + cls slicing +
    clamp + softmax + graph build in one kernel, no host synchronisation.
    """

    #: replay-cache size: one captured hipGraph per distinct set of backbone tap buffers (see `forward`)
    max_graphs = 8

    def __init__(self, ingredient_wrapper: nn.Module, schema_net: SchemaNet, matcher: Matcher):
        super().__init__()
        self.ingredient_wrapper = ingredient_wrapper
        self.schema_net = schema_net
        self.matcher = matcher
        self.num_classes = schema_net.num_classes
        # eval() under torch.no_grad(): the part of `forward` behind the backbone (S1, instance graph, matcher: ~20
        # short launches, host-bound when launched from Python) is captured into a hipGraph per tap-buffer set and
        # replayed ("the API path").  SN_PREDICTOR_GRAPH=0 or `graph_replay = False`: eager launches.
        self.graph_replay = os.environ.get("SN_PREDICTOR_GRAPH", "1") != "0"
        # `output_ring = True` (SN_PREDICTOR_RING=1): `forward` hands out `pred` from a two-deep ring - two captures per
        # tap-buffer set, no copy kernel between two graph launches - valid until the next call BUT ONE (see
        # `_forward_replayed`).  Off by default: the reference's forward returns a tensor the caller owns for ever, and an
        # evaluation loop that collects `output["pred"]` over the batches would read rewritten buffers.
        self.output_ring = os.environ.get("SN_PREDICTOR_RING", "0") == "1"
        # eval() under no_grad: `class_edges` of the returned dictionary is computed when it is read (LazyOutputs);
        # SN_PREDICTOR_LAZY_EDGES=0 / `lazy_class_edges = False`: written by every forward, like the reference
        self.lazy_class_edges = os.environ.get("SN_PREDICTOR_LAZY_EDGES", "1") != "0"
        self._graphs = collections.OrderedDict()
        self._graph_misses = 0                                # CONSECUTIVE calls that found nothing to replay
        self._key_dicts_cache = None
        self._streams = None                                  # `predict_batches`: one HIP stream per batch in flight

    def train(self, mode: bool = True):
        """eval(): the class-graph features (a function of parameters only) are cached across forwards; train(): off"""
        super().train(mode)
        self.matcher.cache_atlas = not mode
        if mode:
            self.matcher.invalidate_atlas_cache()
            self._graphs.clear()
        self._graph_misses = 0                                # (a fine-tune loop alternates train / eval: every eval phase starts afresh)
        return self

    def invalidate_graphs(self):
        """Forget the captured launch sequences (needed only after writes that bypass the tensors' version counters,
        e.g. `p.data.copy_`, or after changing a scalar option of `schema_net` / `matcher`)."""
        self._graphs.clear()
        self._graph_misses = 0
        self._key_dicts_cache = None

    # ---- the path behind the backbone -----------------------------------------------------------------------
    def _after_backbone(self, output, requires_graph: bool, side_stream=None):
        ret = LazyOutputs()
        if self._trains():
            # Training (round 6: the reference's own call sequence is the fast path - worker_schema_net.py:121-147 calls
            # `self.predictor(x)` between `normalize()` and the loss): the route `train.GraphedTrainIter` captures and the bench's
            # eager leg times - the differentiable atlas (`class_edges` with its gradient: what the loss's entropy terms read), the
            # padded instance batch without a host synchronisation, `Matcher.forward_padded` with ONE folded embedding table for
            # both GNN passes and the instance pass on a second stream beside the class pass.  (Until round 5 a training forward
            # went through the inference fork - class branch on the side stream, the table folded once per pass.)
            atlas = self.schema_net.get_atlas()
            graph = self.schema_net.instance_graph_padded(output["ingredients"], output["attn"], output["attn_cls"],
                                                          zero_padding=requires_graph, return_attn_cls=requires_graph,
                                                          rerank=output.get("rerank"))
            ret["pred"] = self.matcher.forward_padded(graph, atlas)
            for k in ("class_vertices", "class_edges", "class_ingredients"):
                ret[k] = atlas[k]
            return self._with_graphs(ret, graph, output) if requires_graph else ret
        # class branch (atlas normalisation + GNN over the K class graphs) on the side stream,
        # instance branch on the current one; joined inside forward_padded
        get_atlas = self.schema_net.get_atlas
        lazy_edges = False
        if not torch.is_grad_enabled() and self.matcher.gnn.masks_adjacency(self.schema_net.edge_weights.tensor):
            # no autograd, MFMA GNN: the GCN operand straight from the pruned parameters.  `class_edges` - nobody's input on
            # this path - from the same pass when the caller is going to look at it (requires_graph, `lazy_class_edges`
            # off), otherwise deferred until read (LazyOutputs: 105 MB less written per call at config [1])
            lazy_edges = self.lazy_class_edges and not requires_graph
            # (a pruned atlas: the operand compacted to the kept vertices of each class - SchemaNet.get_atlas; the fully fused
            # GNN route only, and only where `class_edges` is not wanted from the same pass)
            mode = ("compact" if self.matcher.gnn.embed_dim == 256 else True) if lazy_edges else "with_edges"
            get_atlas = lambda: self.schema_net.get_atlas(fused_adjacency=mode)       # noqa: E731
        # (S1: one call at a time the fp64 finish of the undecided tokens rides in the instance-graph kernel - `rerank` in
        # `output` -, with several batches in flight (`predict_batches`, side_stream False) it stays a launch of its own: the
        # light kernel runs beside the other batches' kernels, the fused form lengthens one that holds every CU - DESIGN 3.1e)
        atlas = self.matcher.atlas_features_async(get_atlas, depends_on=self._atlas_depends_on(), side_stream=side_stream)
        # (the zero padding of the instance edges is only written when the caller asks for the graphs)
        graph = self.schema_net.instance_graph_padded(output["ingredients"], output["attn"], output["attn_cls"],
                                                      zero_padding=requires_graph, return_attn_cls=requires_graph,
                                                      rerank=output.get("rerank"))
        ret["pred"] = self.matcher.forward_padded(graph, atlas.class_dict, feat_kg=atlas)
        for k in ("class_vertices", "class_edges", "class_ingredients"):                  # (the reference's keys, in its order)
            if k == "class_edges" and lazy_edges:
                ret.set_lazy(k, self._class_edges_now)
            else:
                ret[k] = atlas.class_dict[k]
        return self._with_graphs(ret, graph, output) if requires_graph else ret

    def _trains(self) -> bool:
        """autograd is on and some parameter behind the backbone wants a gradient: `forward` is a training forward"""
        if not torch.is_grad_enabled():
            return False
        sn = self.schema_net
        if not sn.vertex_weights.tensor.is_cuda:
            return False
        return any(p.requires_grad for p in sn.parameters()) or any(p.requires_grad for p in self.matcher.parameters())

    @staticmethod
    def _with_graphs(ret, graph, output):
        """requires_graph: the instance graphs as the reference returns them after Matcher's in-place padding (one host read)"""
        n = int(graph["n_max"].item())
        bs = graph["ids"].shape[0]
        ret["instance_ingredients"] = [graph["ids"][b, :n] for b in range(bs)]
        ret["instance_vertices"] = [graph["vertices"][b, :n] for b in range(bs)]
        ret["instance_edges"] = [graph["edges"][b, :n, :n] for b in range(bs)]
        ret["ingredients"] = output["ingredients"]
        ret["attn_cls"] = graph["attn_cls"]          # [bs, L] head mean, clamp-masked (reference schema_net.py:296)
        return ret

    def _class_edges_now(self):
        """`class_edges` of the reference's dictionary from the parameters as they are now: one fused HIP pass
        (reference schema_net.py:152-175: prune, clamp, row-normalise; w / row sum, the reference's own form)"""
        from cpp_extension import ops
        sn = self.schema_net
        with torch.no_grad():
            return ops.atlas_normalize(sn.vertex_weights.tensor.detach(), sn.edge_weights.tensor.detach(), sn.prune_node_threshold,
                                       sn.remove_self_loop)[1]

    def _atlas_depends_on(self):
        """what the cached class-graph features are a function of besides the GNN weights (Matcher.cache_atlas)"""
        sn = self.schema_net
        return (sn.vertex_weights.tensor, sn.edge_weights.tensor, sn.class_ingredients.tensor,
                ("prune", sn.prune_node_threshold, "self_loop", sn.remove_self_loop))

    def _key_dicts(self):
        """The `_parameters` / `_buffers` dicts of every module behind the backbone, collected once: `_replay_key` reads
        their CURRENT values on every call (a replaced Parameter is seen), without the recursion, name building and
        de-duplication of `Module.parameters()` - that walk was 40-50 us of host time per call.  Adding or removing
        SUB-MODULES after the first call is not seen: `invalidate_graphs()`."""
        if self._key_dicts_cache is None:
            mods = [self.schema_net, self.matcher]
            disc = getattr(self.ingredient_wrapper, "discretization_jit", None)     # (the backbone's own weights do not enter the captured part)
            if disc is not None:
                mods.append(disc)
            dicts = []
            for m_ in mods:
                for sub in m_.modules():
                    if sub._parameters:
                        dicts.append(sub._parameters)
                    if sub._buffers:
                        dicts.append(sub._buffers)
            self._key_dicts_cache = dicts
        return self._key_dicts_cache

    def _replay_key(self, mid_feat, extracted, side_stream=None):
        """A capture reads its inputs and every parameter BY ADDRESS and bakes in the operands derived from parameters
        (packed codebook, GNN.prepare, the cached class-graph features): it stays valid while the tap buffers are the
        same memory (the caching allocator hands a steady inference loop the same blocks every iteration) and no
        parameter / buffer of the modules behind the backbone has been written or moved.  Version counters only ever
        grow, so their SUM over a fixed set of tensors changes with every write; the addresses are summed with distinct
        odd weights (a swap of two buffers changes the sum)."""
        ver = 0
        ptr = 0
        i = 1
        for d in self._key_dicts():
            for t in d.values():
                if t is not None:
                    ver += t._version
                    ptr += i * t.data_ptr()
                    i += 2
        sn = self.schema_net
        # a capture owns ONE set of buffers: it must never run on two streams at once (`predict_batches`)
        return (mid_feat.data_ptr(), extracted.data_ptr(), mid_feat.shape, extracted.shape, mid_feat.stride(), extracted.stride(),
                mid_feat.dtype, extracted.dtype, ver, ptr, i, self.matcher.cache_atlas, sn.prune_node_threshold, sn.remove_self_loop,
                sn.clamp_vertex_attn, sn.clamp_edge_attn, side_stream, torch.cuda.current_stream().cuda_stream)

    def _give_up_replay(self, why: str):
        self.schema_net.logger.warning("SchemaNetPredictor: hipGraph replay switched off (%s): eager launches from now on; "
                                       "`graph_replay = True` switches it back on", why)
        self.graph_replay = False
        self._graphs.clear()
        self._graph_misses = 0

    def _forward_replayed(self, x, own_pred: bool = True, side_stream=None):
        """eval + no_grad + `taps`: backbone eagerly, then the captured launch sequence of everything behind it.

        own_pred: `pred` must survive the next call.  The capture's output buffer is rewritten by its next replay: by default
        the caller gets a copy (`clone()`: its own tensor, as from the reference).  With `output_ring` a
        key gets a SECOND capture of the same step (with buffers of its own: a shared memory pool would let one capture's
        intermediates land on the other's outputs) and the calls alternate between the two: the tensor a
        call returns is valid until the next call BUT ONE with the same taps - a two-deep output ring instead of a copy
        kernel between two graph launches."""
        from ..utils.graph_replay import GraphedStep
        wrapper = self.ingredient_wrapper
        out_backbone = wrapper.backbone_jit(x)
        mid_feat, extracted = out_backbone["mid_feat"], out_backbone["extracted"]
        key = self._replay_key(mid_feat, extracted, side_stream)
        entry = self._graphs.get(key)
        ring = own_pred and self.output_ring
        if entry is None or (ring and len(entry) < 3 and entry[0] == 1):
            # consecutive misses: tap buffers that keep moving (or weights written before every call) - replay cannot pay
            if entry is None:
                if self._graph_misses >= 4 * self.max_graphs:
                    self._give_up_replay(f"{self._graph_misses} consecutive calls found no capture to replay: the backbone's tap "
                                         "buffers or the parameters change on every call")
                    return self._after_backbone(wrapper.taps_from(out_backbone, defer=side_stream is not False), False, side_stream)
                self._graph_misses += 1
            try:
                step = GraphedStep(lambda: self._after_backbone(wrapper.taps_from(out_backbone, defer=side_stream is not False), False, side_stream))
            except Exception as exc:                              # noqa: BLE001 - a configuration that cannot be captured
                self._give_up_replay(f"capture failed: {exc!r}")
                return self._after_backbone(wrapper.taps_from(out_backbone, defer=side_stream is not False), False, side_stream)
            if entry is None:
                entry = [0, step]                                 # [calls so far, capture A (, capture B)]
                self._graphs[key] = entry
                while len(self._graphs) > self.max_graphs:
                    self._graphs.popitem(last=False)
            else:
                entry.append(step)
            # (no reference to the tap tensors is kept: a backbone that allocates its outputs afresh gets the SAME blocks back
            # from the caching allocator once the previous iteration's are released - that is what makes the next call a
            # hit.  The capture is only ever replayed when both taps live at the captured addresses again.)
        else:
            self._graph_misses = 0
            self._graphs.move_to_end(key)
        step = entry[1 + (entry[0] & 1)] if (ring and len(entry) > 2) else entry[1]
        entry[0] += 1
        ret = step.replay().lazy_copy()
        if own_pred and not self.output_ring:
            ret["pred"] = ret["pred"].clone()                      # the capture's own buffer is rewritten by the next replay
        return ret

    def predict_batches(self, batches, depth: int = 4):
        """`forward(x)` for every `x` of an iterable (an evaluation loop: reference eval/evaluation.py:63-80 calls the model
        once per batch of the loader), yielded in order, with up to `depth` batches in flight on `depth` HIP streams.

        One step of this path is a chain of ~30 short kernels, most of which fill every CU's LDS on their own: one batch
        at a time leaves the chip idle at every kernel boundary.  Batch i runs - backbone eagerly, the rest replayed from
        the capture of its stream - on stream i % depth; its results are handed to the caller's stream (an event wait,
        no host synchronisation) when batch i + depth is submitted or the iterable ends.  `pred` is the batch's own
        tensor; `class_*` are those of the capture (functions of the parameters only).  Same conditions as the replayed
        `forward` (eval, no autograd, a wrapper with taps, a GPU); otherwise the batches are simply run one by one."""
        wrapper = self.ingredient_wrapper
        if not (self.graph_replay and not self.training and not torch.is_grad_enabled() and hasattr(wrapper, "taps_from")
                and self.schema_net.vertex_weights.tensor.is_cuda and not torch.cuda.is_current_stream_capturing()):
            for x in batches:
                yield self.forward(x)
            return
        depth = max(1, min(int(depth), self.max_graphs))
        if self._streams is None or len(self._streams) < depth:
            self._streams = [torch.cuda.Stream() for _ in range(depth)]
        home = torch.cuda.current_stream()
        in_flight = collections.deque()
        # one batch at a time: the class-graph branch overlaps the instance chain on a second stream; several in flight:
        # every batch in line on its own stream (the other batches fill the gaps; forks on top of that only contend - DESIGN 5)
        fork = None if depth == 1 else False

        def hand_over(item):
            # the capture's `pred` buffer is copied on the CALLER's stream: nothing allocated on a batch stream outlives
            # the call that made it, so the allocator hands every visit of a stream the same tap blocks (= replay hits);
            # the next replay of that capture is submitted after this copy (`st.wait_stream(home)` below)
            out, done = item
            home.wait_event(done)
            out["pred"] = out["pred"].clone()
            return out

        for i, x in enumerate(batches):
            if len(in_flight) == depth:
                yield hand_over(in_flight.popleft())
            st = self._streams[i % depth]
            st.wait_stream(home)                       # `x` (and anything the parameters wait for) was produced there
            with torch.cuda.stream(st):
                if torch.is_tensor(x):
                    x.record_stream(st)
                if self.graph_replay:
                    out = self._forward_replayed(x, own_pred=False, side_stream=fork)
                else:                                  # (capture gave up on the way: eager launches, still `depth` streams)
                    out = self._after_backbone(wrapper.taps(x, defer=fork is not False), False, fork)
                    out["pred"].record_stream(home)
                done = torch.cuda.Event()
                done.record(st)
            in_flight.append((out, done))
        while in_flight:
            yield hand_over(in_flight.popleft())

    def forward(self, x: torch.Tensor, requires_graph: bool = False) -> Dict[str, torch.Tensor]:
        wrapper = self.ingredient_wrapper
        if (self.graph_replay and not self.training and not requires_graph and not torch.is_grad_enabled()
                and hasattr(wrapper, "taps_from") and self.schema_net.vertex_weights.tensor.is_cuda
                and not torch.cuda.is_current_stream_capturing()):       # (inside somebody else's capture: plain launches, theirs to replay)
            return self._forward_replayed(x)
        with torch.no_grad():
            if hasattr(wrapper, "taps"):
                output = wrapper.taps(x, defer=True)
            else:
                output = wrapper(x)
        return self._after_backbone(output, requires_graph)
