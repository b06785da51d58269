"""Frame-list sharding for multi-GPU runs: frames are independent, so a batch is split into
contiguous chunks, one per rank (one process per GPU), with no collective on the data path."""

from __future__ import annotations


def shard_range(num_frames: int, rank: int, world_size: int) -> range:
    """Indices of the frames rank `rank` processes.  Contiguous chunks; the first
    `num_frames % world_size` ranks get one extra frame."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of {world_size}')
    base, extra = divmod(num_frames, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_frames(frames: list, rank: int, world_size: int) -> list:
    return [frames[i] for i in shard_range(len(frames), rank, world_size)]


class FrameStreams:
    """Runs a per-frame chain over the frames of a batch round-robin on `streams` HIP streams of one GPU.

    Frames are independent, and a chain of stencil kernels leaves the GPU partly idle at every kernel's tail (a last,
    partially filled round of workgroups; one-workgroup finish kernels; launch gaps): with a second stream the next
    frame's kernels fill those gaps (+15 % on the 12 MP RCD -> Wiener -> bilateral -> Reinhard chain).  Every stream gets
    its OWN chain object from `make_chain()` (op workspaces, accumulators, hand-over planes), so nothing is shared
    between frames in flight; the results are bit-identical to running the frames back to back.

        runner = FrameStreams(device, make_chain, streams=2)
        outputs = runner.run(frames)      # list, in frame order; safe to use on the caller's current stream
    """

    def __init__(self, device, make_chain, streams: int = 2):
        import torch

        if streams < 1:
            raise ValueError('streams must be >= 1')
        self.device = device
        self.chains = [make_chain() for _ in range(streams)]
        self.streams = [torch.cuda.Stream(device) for _ in range(streams)] if streams > 1 else [None]
        self._next = 0  # the stream the next frame goes to: the rotation carries over from batch to batch
        self._unjoined = False  # frames issued since the last join()

    def issue(self, frames) -> list:
        """Launch every frame's chain; returns the outputs without joining the streams (they are complete only after
        `join()` or a device synchronisation)."""
        import torch

        outs = []
        if self.streams[0] is None:
            return [self.chains[0](f) for f in frames]
        here = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(here)  # the inputs were produced on the caller's stream
        from .torch_darktable_extension import concurrent_frames

        n = len(self.streams)
        # a single frame with nothing else in flight would run the register-blocked RCD strips ALONE, where they are the slower
        # variant (csrc/tdk_rcd_quad.h): tell the ops about company only when this batch (or an un-joined earlier one) provides it
        company = len(frames) > 1 or self._unjoined
        self._unjoined = True
        with concurrent_frames(company):  # the other streams' frames share the GPU with every kernel launched here
            for f in frames:
                k = self._next
                self._next = (k + 1) % n
                with torch.cuda.stream(self.streams[k]):
                    outs.append(self.chains[k](f))
        return outs

    def join(self, outs=()) -> None:
        """Make the caller's current stream wait for all frame streams; `outs` are marked as used on it (they were
        allocated on the frame streams)."""
        import torch

        if self.streams[0] is None:
            return
        here = torch.cuda.current_stream(self.device)
        for s in self.streams:
            here.wait_stream(s)
        for o in outs:
            if isinstance(o, torch.Tensor):
                o.record_stream(here)
        self._unjoined = False

    def run(self, frames) -> list:
        outs = self.issue(frames)
        self.join(outs)
        return outs

    def capture(self, frames) -> 'CapturedBatch':
        """One batch -- every frame's chain on its stream, fork and join included -- captured into ONE HIP graph
        (torch.cuda.CUDAGraph).  `frames` are the STATIC input tensors: refill them in place (copy_) and call replay() for every
        further batch; the outputs are static tensors as well (overwritten by the next replay).  Nothing in the library allocates,
        synchronises or keeps process-global state inside an op, so the whole chain is capturable; what a replay saves is the
        host's launch path (57 ctypes calls per batch of 8 frames -> one graph launch).  The chains must have run once on their
        streams before (workspaces allocated, LDS limits raised): capture() does that warm-up itself."""
        import torch

        for _ in range(2):  # warm-up outside the capture: workspace allocation, hipFuncSetAttribute, cached scalars
            self.run(frames)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = self.run(frames)
        return CapturedBatch(graph, list(frames), outs)


class CapturedBatch:
    """A FrameStreams batch as a replayable HIP graph: `inputs` and `outputs` are the static tensors of the capture."""

    def __init__(self, graph, inputs, outputs):
        self.graph, self.inputs, self.outputs = graph, inputs, outputs

    def replay(self, new_inputs=None) -> list:
        if new_inputs is not None:
            for dst, src in zip(self.inputs, new_inputs):
                dst.copy_(src)
        self.graph.replay()
        return self.outputs
