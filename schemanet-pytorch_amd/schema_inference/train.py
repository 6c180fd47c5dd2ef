"""The training step of the reference's SchemaNet trainer as a function (reference
schema_inference/tasks/worker_schema_net.py:121-147 `train_iter`, :371-378 parameter groups), so that config 5
(`4.train_schema_net.sh`) can be driven without the reference's launcher / logging / checkpoint loop:

    optimizer.zero_grad(); schema_net.normalize(); output = predictor(x)
    loss = sum(loss_weights[k.split(".")[0]] * v for k, v in loss_fn(output, targets).items() if the prefix has a weight)
    loss.backward(); optimizer.step(); optimizer.zero_grad(set_to_none=True)

With autograd enabled the modules of this package run their differentiable route (HIP forward of S1 and of the
instance graph with gradients to the attribute weights, torch ops for the GCN); the sparsity terms of the loss use
the HIP row-entropy kernels."""
import re
from collections import OrderedDict
from typing import Any, Callable, Dict, Iterable, List, Tuple

import torch
import torch.nn as nn


def param_groups(named_parameters: Iterable[Tuple[str, nn.Parameter]], groups: List[Dict[str, Any]],
                 drop_remain: bool = False) -> List[Dict[str, Any]]:
    """reference utils/customs_param_group.py: every group takes the not-yet-taken parameters whose name matches its
    regular expression (re.match, sorted names) and carries its `cfg` as optimizer options; the rest form a last
    group unless `drop_remain`, in which case they are set to requires_grad False."""
    left = OrderedDict(named_parameters)
    out = []
    for group in groups:
        names = [n for n in sorted(left) if re.match(group["pattern"], n)]
        if not names:
            raise AssertionError("no matched for pattern {}".format(group["pattern"]))
        out.append(dict(params=[left.pop(n) for n in names], **group.get("cfg", dict())))
    if left and not drop_remain:
        out.append(dict(params=list(left.values())))
    elif drop_remain:
        for p in left.values():                 # like the reference: what no group took is frozen
            p.requires_grad_(False)
    return out


def weighted_total(loss_dict: Dict[str, torch.Tensor], loss_weights: Dict[str, float]) -> torch.Tensor:
    return sum(v * loss_weights[k.split(".")[0]] for k, v in loss_dict.items() if k.split(".")[0] in loss_weights)


def train_iter(forward: Callable[[], Dict[str, torch.Tensor]], schema_net, loss_fn, loss_weights: Dict[str, float],
               optimizer: torch.optim.Optimizer, targets: Dict[str, torch.Tensor], scaler=None):
    """One optimisation step.  `forward()` runs the predictor on the batch (e.g. `lambda: predictor(x)`) and returns
    its output dictionary (`pred`, `class_vertices`, `class_edges`, ...).  -> (total loss (detached), loss dict).
    scaler: a `torch.amp.GradScaler` = the reference's `use_amp: True` route (worker_schema_net.py:128-143): forward and loss
    under `torch.autocast`, the scaled loss back-propagated, `scaler.step` / `scaler.update`.  The HIP-backed ops of this
    package keep computing in fp32 under autocast (cpp_extension/ops.py: custom_fwd(cast_inputs=torch.float32))."""
    optimizer.zero_grad()
    schema_net.normalize()
    with torch.autocast("cuda", enabled=scaler is not None):
        output = forward()
        loss_dict = loss_fn(output, targets)
        loss = weighted_total(loss_dict, loss_weights)
    if scaler is not None:
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
    else:
        loss.backward()
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return loss.detach(), OrderedDict((k, v.detach()) for k, v in loss_dict.items())


class GraphedTrainIter:
    """`train_iter` for one batch shape as ONE hipGraph replay: `normalize -> forward -> loss -> backward -> optimizer.step` are
    captured once (PyTorch's whole-network capture) and every later call copies the batch into the static input buffers
    and replays - no python, no ~300 launches, no launch gaps between the many small kernels of the instance side.

        step = GraphedTrainIter(lambda batch: predictor_part(batch), schema_net, loss_fn, loss_weights, optimizer, batch, targets)
        loss, loss_dict = step(batch, targets)          # static tensors: valid until the next call

    Conditions (checked where they can be): a CUDA optimizer built with `capturable=True` (e.g.
    `torch.optim.AdamW(params, ..., capturable=True, fused=True)`); a `forward` free of host synchronisations - the padded
    route `SchemaNet.instance_graph_padded` + `Matcher.forward_padded` that `SchemaNetPredictor` takes, not the reference's
    python lists; fixed shapes and dtypes of every tensor in `batch` / `targets`.  `warmup` eager iterations run first on a
    side stream (library workspaces, the optimizer's state): they are REAL optimisation steps on the example batch, as in
    the PyTorch recipe - the step count of the trajectory includes them.
    The values are those of `train_iter` (the same kernels in the same order)."""

    def __init__(self, forward: Callable[[Dict[str, torch.Tensor]], Dict[str, torch.Tensor]], schema_net, loss_fn,
                 loss_weights: Dict[str, float], optimizer: torch.optim.Optimizer, batch: Dict[str, torch.Tensor],
                 targets: Dict[str, torch.Tensor], warmup: int = 2, compact: bool = True):
        # compact: the class GNN of the captured iteration runs on the kept vertices of a pruned IR-Atlas where that pays
        # (`schema_net.compact_training`, SchemaNet.get_atlas): the extra launches of that route cost a replay nothing
        # The switch is a property of THIS captured iteration: it is set for the warm-up and the capture and put back
        # afterwards, so that eager `train_iter` calls on the same SchemaNet keep their own (uncompacted, faster eagerly) route.
        # `self.compacted` says which route the graph holds: the decision is taken at the warm-up (kept fraction under 3/4) and is
        # baked into the replay - build a new GraphedTrainIter when the atlas has changed enough for the other route to pay.
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainIter needs an optimizer built with capturable=True")
        if warmup < 1 and not optimizer.state:
            raise ValueError("GraphedTrainIter: the optimizer has no state yet - at least one warm-up iteration is needed")
        had_switch = getattr(schema_net, "compact_training", False)
        if compact and hasattr(schema_net, "get_atlas"):
            schema_net.compact_training = True
        self.batch = {k: v.clone() for k, v in batch.items()}
        self.targets = {k: v.clone() for k, v in targets.items()}
        self.optimizer, self.schema_net = optimizer, schema_net

        def iteration():
            schema_net.normalize()
            output = forward(self.batch)
            loss_dict = loss_fn(output, self.targets)
            loss = weighted_total(loss_dict, loss_weights)
            loss.backward()
            optimizer.step()
            return loss.detach(), OrderedDict((k, v.detach()) for k, v in loss_dict.items())

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                iteration()
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad(set_to_none=True)                   # (the captured backward allocates the .grad tensors in the graph's pool)
        # keep_graph: the captured hipGraph_t is post-processed before it is instantiated - the memset nodes the library's own
        # ops leave in it (semaphores of multi-block reductions, embedding_dense_backward's zero fill) become kernel nodes
        # (sn_graph_replace_memsets: a captured memset node is not reliable on ROCm 7.2; seen here as reductions that keep the
        # previous replay's result, i.e. a trajectory that drifts from the second replay on)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(self.graph):
            self.loss, self.loss_dict = iteration()
        from ctypes import byref, c_int, c_void_p
        from cpp_extension import _native as N
        done, left = c_int(0), c_int(0)
        N.check(N.require_gpu().sn_graph_replace_memsets(c_void_p(self.graph.raw_cuda_graph()), byref(done), byref(left)), "sn_graph_replace_memsets")
        self.memsets_replaced, self.memsets_left = done.value, left.value
        if self.memsets_left:
            import warnings
            warnings.warn(f"{self.memsets_left} memset node(s) of the captured graph could not be replaced by kernel nodes (two-dimensional "
                          "memsets): on ROCm 7.2 a captured memset node may not clear on replay - compare a replay with an eager call")
        self.graph.instantiate()
        self.warmup_steps = warmup
        state = getattr(schema_net, "_compaction_state", None)
        self.compacted = bool(compact and state is not None and state.get("decision", False))
        if compact and hasattr(schema_net, "get_atlas"):
            schema_net.compact_training = had_switch

    def __call__(self, batch: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor]):
        for mine, theirs in ((self.batch, batch), (self.targets, targets)):
            for k, v in mine.items():
                if theirs[k] is not v:
                    v.copy_(theirs[k], non_blocking=True)
        self.graph.replay()
        return self.loss, self.loss_dict
