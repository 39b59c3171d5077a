"""hipGraph capture of the encoders (static shapes / graph capture of the step's launch-bound parts).

The CLIP ViT and the AST encoder have no data-dependent shape and no host-built tensor inside: per step and encoder the host issues
~320 ctypes launches forward and ~560 backward through as many autograd nodes (tools/host_profile.py). `GraphedSegment` turns one such
encoder into two hipGraphs -- forward, and backward incl. the gradient-arena writes of its wgrad kernels -- and a single autograd node
that replays them:

  * capture happens INSIDE a training step, on the `warmup`-th call with a given (shape, dtype) key: the calls before it ran eagerly
    (first-use attribute settings, workspaces, cached index tensors all exist), the capturing call replays what it just captured;
  * forward capture runs the eager function on a static input buffer under torch.cuda.graph (private memory pool: the activations
    saved for backward live there and are reused by every replay); backward capture runs torch.autograd.backward on the captured
    output against a static gradient buffer in the same pool (what torch.cuda.make_graphed_callables does for nn.Modules);
  * parameter gradients never reach autograd on this path (ops.GradSink: kernels accumulate into the flat arena), so the data-parallel
    reducer's bookkeeping -- host code that runs beside the launches -- would be skipped by a replay: the names reported during capture
    are recorded and reported again after every backward replay, in the same order (the capture-time bucket schedule of DESIGN 6);
  * dropout: by-value (seed, offset) arguments are baked into the graph; ops.DropoutState's device mode restarts the by-value offsets
    every step and adds a device-resident per-step counter inside the kernels (include/valor_hip.h `rng_base`), so replays draw
    fresh masks and an eager run with the same seed draws the SAME masks (bit-identical losses, tests/test_graphs_gpu.py);
  * a replay runs on the caller's current stream (the AST encoder's graphs replay on the encoders' side stream), static buffers are
    only touched in stream order.
The reference has no counterpart (eager PyTorch, model/pretrain.py:246-263; apex amp O2 around it, train_utils.py:226-232)."""
import torch
from torch.autograd import Function

from . import ops


class _Replay(Function):
    @staticmethod
    def forward(ctx, seg, anchor, x):
        if x.data_ptr() != seg.static_in.data_ptr():
            seg.static_in.copy_(x)
        seg.g_fwd.replay()
        ops.DropoutState.offset += seg.draws
        ctx.seg = seg
        return seg.static_out.view_as(seg.static_out)

    @staticmethod
    def backward(ctx, g):
        seg = ctx.seg
        seg.static_gout.copy_(g)
        seg.g_bwd.replay()
        if ops.GradSink.listener is not None:
            for name in seg.sunk:
                ops.GradSink.listener(name)
        return None, None, None


class _Captured:
    __slots__ = ("g_fwd", "g_bwd", "static_in", "static_out", "static_gout", "sunk", "draws", "offset0", "stream")


class GraphedSegment:
    """fn: device tensor -> device tensor (an encoder); anchor: any parameter of it that requires grad (gives the replay node a
    differentiable input so autograd calls its backward)."""

    def __init__(self, name, fn, anchor, warmup=2):
        self.name, self.fn, self.anchor, self.warmup = name, fn, anchor, warmup
        self.calls = {}
        self.captured = {}

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, ops.DropoutState.offset)
        cap = self.captured.get(key)
        if cap is None:
            n = self.calls.get(key, 0)
            self.calls[key] = n + 1
            if n < self.warmup:
                return self.fn(x)
            cap = self._capture(x)
            self.captured[key] = cap
        return _Replay.apply(cap, self.anchor, x)

    def _capture(self, x):
        cap = _Captured()
        cap.stream = torch.cuda.Stream(device=x.device)
        cap.static_in = x.detach().clone()
        cap.offset0 = ops.DropoutState.offset
        cap.g_fwd, cap.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(cap.g_fwd, stream=cap.stream):
            out = self.fn(cap.static_in)
        cap.draws = ops.DropoutState.offset - cap.offset0
        ops.DropoutState.offset = cap.offset0                  # the replay of this very call advances it again
        cap.static_out = out
        cap.static_gout = torch.zeros_like(out)
        rec = []
        ops.GradSink.recorder = rec
        try:
            with torch.cuda.graph(cap.g_bwd, pool=cap.g_fwd.pool(), stream=cap.stream):
                torch.autograd.backward((out,), (cap.static_gout,))
        finally:
            ops.GradSink.recorder = None
        cap.sunk = rec
        return cap

    def release(self):
        self.captured.clear()
        self.calls.clear()
