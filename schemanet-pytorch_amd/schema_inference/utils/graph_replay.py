"""hipGraph replay of an inference step (no reference counterpart: the reference launches every op
eagerly from Python; on MI355X the schema-inference step is ~30 short kernels on two HIP streams,
so the launch path is captured once and replayed).

Everything in the step runs without a host synchronisation (device-side extents, fixed-shape
padded outputs, the side-stream class branch joined by an event), which is what makes it
capturable.  `GraphedStep(fn)` warms `fn` up on a capture stream, records ONE call of it into a
`torch.cuda.CUDAGraph` (a hipGraph on ROCm) and `replay()` re-runs the recorded kernels on the
same buffers: inputs are read from the tensors `fn` closed over at capture time, so update them in
place (`tensor.copy_`) between replays; the outputs are the tensors returned at capture time."""
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
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs


class PipelinedSteps:
    """`depth` independent captures of the same step, replayed round-robin on `depth` HIP streams: batch i+1
    starts while batch i is still in its tail (most kernels of the step fill every CU's LDS on their own, so one
    graph alone leaves the chip idle at every kernel boundary and during the narrow kernels).  Every capture has
    its own buffers; `fn` must only share read-only state and commutative accumulators (atomic adds) between
    calls.  `submit()` returns the outputs of the capture it replayed - valid after `join()` or after the next
    `submit()` on the same slot has been ordered behind a reader."""

    def __init__(self, fn, depth: int = 2):
        self.steps = [GraphedStep(fn) for _ in range(depth)]
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self.i = 0

    def submit(self):
        k = self.i % len(self.steps)
        self.i += 1
        if self.i <= len(self.steps):                        # first use of the slot: behind whatever produced the inputs
            self.streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.streams[k]):
            self.steps[k].graph.replay()
        return self.steps[k].outputs

    def join(self):
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)
