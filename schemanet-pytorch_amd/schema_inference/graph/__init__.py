"""`schema_inference.graph` -- the reference's public names (graph/__init__.py:7-9, 14-57),
MI355X-native.  `to_networkx` (visualisation helper) is outside the hot path and not provided.
"""
import collections
from typing import Dict

import torch
import torch.nn as nn

from .schema_net import SchemaNet
from .match import Matcher
from .gnn import GNN
from .statistics import SchemaStatistics, shard_indices

__all__ = ["SchemaNet", "Matcher", "GNN", "SchemaNetPredictor", "SchemaStatistics", "shard_indices"]


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

    def __init__(self, ingredient_wrapper: nn.Module, schema_net: SchemaNet, matcher: Matcher):
        super().__init__()
        self.ingredient_wrapper = ingredient_wrapper
        self.schema_net = schema_net
        self.matcher = matcher
        self.num_classes = schema_net.num_classes

    def train(self, mode: bool = True):
        """eval(): the class-graph features (a function of parameters only) are cached across forwards; train(): off"""
        super().train(mode)
        self.matcher.cache_atlas = not mode
        if mode:
            self.matcher.invalidate_atlas_cache()
        return self

    def forward(self, x: torch.Tensor, requires_graph: bool = False) -> Dict[str, torch.Tensor]:
        ret = collections.OrderedDict()
        with torch.no_grad():
            if hasattr(self.ingredient_wrapper, "taps"):
                output = self.ingredient_wrapper.taps(x)
            else:
                output = self.ingredient_wrapper(x)
        # class branch (atlas normalisation + GNN over the K class graphs) on the side stream,
        # instance branch on the current one; joined inside forward_padded
        atlas = self.matcher.atlas_features_async(
            self.schema_net.get_atlas, depends_on=(self.schema_net.vertex_weights.tensor, self.schema_net.edge_weights.tensor))
        # (the zero padding of the instance edges is only written when the caller asks for the graphs)
        graph = self.schema_net.instance_graph_padded(output["ingredients"], output["attn"], output["attn_cls"],
                                                      zero_padding=requires_graph, return_attn_cls=requires_graph)
        ret["pred"] = self.matcher.forward_padded(graph, atlas.class_dict, feat_kg=atlas)
        class_dict = atlas.class_dict
        ret.update(class_dict)
        if requires_graph:
            n = int(graph["n_max"].item())
            bs = graph["ids"].shape[0]
            ret["instance_ingredients"] = [graph["ids"][b, :n] for b in range(bs)]
            ret["instance_vertices"] = [graph["vertices"][b, :n] for b in range(bs)]
            ret["instance_edges"] = [graph["edges"][b, :n, :n] for b in range(bs)]
            ret["ingredients"] = output["ingredients"]
            ret["attn_cls"] = graph["attn_cls"]          # [bs, L] head mean, clamp-masked (reference schema_net.py:296)
        return ret
