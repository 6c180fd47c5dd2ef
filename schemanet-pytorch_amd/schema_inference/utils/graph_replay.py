"""hipGraph replay of an inference step (no reference counterpart: the reference launches every op
eagerly from Python; on MI355X the schema-inference step is ~30 short kernels on two HIP streams,
so the launch path is captured once and replayed).

Everything in the step runs without a host synchronisation (device-side extents, fixed-shape
padded outputs, the side-stream class branch joined by an event), which is what makes it
capturable.  `GraphedStep(fn)` warms `fn` up on a capture stream, records ONE call of it into a
`torch.cuda.CUDAGraph` (a hipGraph on ROCm) and `replay()` re-runs the recorded kernels on the
same buffers: inputs are read from the tensors `fn` closed over at capture time, so update them in
place (`tensor.copy_`) between replays; the outputs are the tensors returned at capture time.

Contract on weights: a capture reads every parameter by address (live values: biases, LayerNorm, the
IR-Atlas), but the operands DERIVED from parameters are whatever existed at capture time - the packed
codebook of S1 (`ops.PackedCodebook`), `GNN.prepare()` (the folded embedding table, the W2 planes, fc^T) and,
with `Matcher.cache_atlas`, the cached class-graph features.  After a weight update (optimizer.step,
load_state_dict, any in-place write) a replay would mix those stale operands with live parameters:
RE-CAPTURE after a weight change.  `SchemaNetPredictor.forward` does this by itself (its replay cache is keyed
on the version counters of every parameter and buffer); a hand-made `GraphedStep` must be rebuilt by its owner."""
import torch


class GraphedStep:
    def __init__(self, fn, warmup: int = 2):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU (the HIP path has no CPU fallback)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                 # first-launch work (attributes, packed codebook, workspaces) is not capturable
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # keep_graph: the captured hipGraph_t stays accessible until it is instantiated below - any memset node in it (a
        # `zeros`, a library reduction's semaphores) is replaced by a kernel node first: a captured memset node is not reliable on
        # ROCm 7.2 (csrc/sn_common.h; seen in train.GraphedTrainIter as reductions that kept the previous replay's result)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        # capture_begin / capture_end by hand: the `torch.cuda.graph` context also empties the caching allocator, which
        # hands every LATER eager allocation of the process a new address - and a capture keyed on the addresses of its
        # inputs (SchemaNetPredictor) would then see them move after every other capture
        with torch.no_grad(), torch.cuda.stream(side):
            self.graph.capture_begin()
            try:
                self.outputs = fn()
            except BaseException:
                try:                                      # leave capture mode, but report the step's own error
                    self.graph.capture_end()
                except Exception:                         # noqa: BLE001 - an invalidated capture fails again here
                    pass
                raise
            self.graph.capture_end()
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
        torch.cuda.current_stream().wait_stream(side)

    def replay(self):
        self.graph.replay()
        return self.outputs


class PipelinedSteps:
    """Captures of the step replayed round-robin on `depth` HIP streams: batch i+1 starts while batch i is still in
    its tail (most kernels of the step fill every CU's LDS on their own, so one graph alone leaves the chip idle at
    every kernel boundary and during the narrow kernels).  `fns` is one callable (captured `depth` times) or a list of
    callables, one per batch buffer (e.g. eight closures over eight different input batches): capture j replays on
    stream j % depth, so `depth` different batches are in flight and the captures are visited in rotation.  Every
    capture has its own buffers; the callables must only share read-only state and commutative accumulators (atomic
    adds) between calls.  `submit()` returns the outputs of the capture it replayed - valid after `join()` or after
    the next `submit()` on the same stream has been ordered behind a reader."""

    def __init__(self, fns, depth: int = 2):
        if callable(fns):
            fns = [fns] * depth
        self.steps = [GraphedStep(fn) for fn in fns]
        self.depth = max(1, min(depth, len(self.steps)))
        if len(self.steps) % self.depth != 0:               # a capture must always replay on the same stream (its buffers)
            raise ValueError(f"{len(self.steps)} captures cannot rotate over {self.depth} streams")
        self.streams = [torch.cuda.Stream() for _ in range(self.depth)]
        self.i = 0

    def submit(self):
        j = self.i % len(self.steps)
        k = self.i % self.depth
        self.i += 1
        if self.i <= self.depth:                             # first use of the stream: behind whatever produced the inputs
            self.streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.streams[k]):
            self.steps[j].graph.replay()
        return self.steps[j].outputs

    def join(self):
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)
