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
    def forward(ctx, seg, anchor, *xs):
        for x, st in zip(xs, seg.static_in):
            if x.data_ptr() != st.data_ptr():
                st.detach().copy_(x)
        seg.g_fwd.replay()
        ops.DropoutState.offset += seg.draws
        ctx.seg = seg
        return seg.static_out.view_as(seg.static_out)

    @staticmethod
    def backward(ctx, g):
        seg = ctx.seg
        seg.static_gout.copy_(g)
        seg.g_bwd.replay()
        listener = ops.GradSink.listener
        if listener is not None:
            hooks = []
            for name in seg.sunk:
                if isinstance(name, tuple):                    # ("hook", name): a post-accumulate hook that ran inside the captured backward
                    hooks.append(name[1])
                else:
                    listener(name)
            if hooks and seg.outer_hooked is None:
                # first backward with this capture: whether the OUTER backward hooks these parameters too (they are also used by eager
                # code of the step) is only known at its end -- report what it did not hook then, and remember the answer
                from torch.autograd.variable import Variable

                def settle(seg=seg, hooks=hooks, listener=listener):
                    seg.outer_hooked = {n for n in hooks if n in ops.GradSink.live_hooks}
                    for n in hooks:
                        if n not in seg.outer_hooked:
                            listener(n)
                Variable._execution_engine.queue_callback(settle)
            else:
                for n in hooks:
                    if n not in seg.outer_hooked:
                        listener(n)
        # differentiable inputs (the decoder stack's token rows and its [video | audio] input): the captured backward left their gradients
        # in static buffers; they are consumed by the nodes behind this one in stream order, before the next replay rewrites them
        return (None, None) + tuple(st.grad if st.requires_grad else None for st in seg.static_in)


class _Captured:
    __slots__ = ("g_fwd", "g_bwd", "static_in", "static_out", "static_gout", "sunk", "draws", "offset0", "stream", "outer_hooked")


# One capture stream per (device, REPLAY stream), shared by every capture that is replayed there. Kernel workspaces and the
# grouped-reduction queue are keyed by stream (kernels.workspace / ReduceQueue): a stream per capture allocated 256 MiB + 1 GiB of them per
# captured key, inside that key's pool. They are scratch inside one kernel sequence, and graphs replayed on ONE stream run one after the
# other: sharing is safe. Graphs that replay on DIFFERENT streams (the ViT on the step's stream, the AST / CLIP text encoders on the side
# stream) run concurrently: they get their own capture stream, hence their own workspaces.
# The private memory POOL is per segment (shared by its shape keys: only one key of a segment is alive in a step). Two segments must not
# share one: a pool hands the memory a finished capture freed -- the saved activations of segment A, consumed by A's captured backward --
# to the next capture, and at replay A's backward runs AFTER B's forward (forward A, forward B, backward B, backward A).
_CAPTURE_CTX = {}


def _capture_ctx(device):
    cur = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), cur.cuda_stream)
    ctx = _CAPTURE_CTX.get(key)
    if ctx is None:
        ctx = _CAPTURE_CTX[key] = {"stream": torch.cuda.Stream(device=device)}
    return ctx


def release_all():
    """drop the shared capture streams and the kernel workspaces that were allocated on them (call after every segment's release())"""
    from . import kernels as K
    for ctx in _CAPTURE_CTX.values():
        K.release_stream(ctx["stream"])
    _CAPTURE_CTX.clear()


class GraphedSegment:
    """fn: device tensor(s) -> device tensor (an encoder: inputs that do not need gradients -- pixels, spectrograms, token ids, masks; the
    decoder stack: its token rows and its [video | audio] input DO, and get their gradients back from static buffers the captured backward
    fills). Every call copies the inputs into the static buffers the graph was captured on. The replay node needs one differentiable
    input for autograd to call its backward even when no real input has one: a private one-element leaf -- NOT a parameter of the model
    (autograd runs a leaf's post-accumulate hooks even when the node hands it no gradient, and the data-parallel reducer counts those
    calls per parameter).
    fn may fork work onto other streams (the decoder's K|V projections run on the encoders' side stream): a stream that waits for the
    capturing stream joins the capture, its work becomes a parallel branch of the graph, and it must be joined back before fn returns
    (every decoder layer waits for its own projection's event).
    One call in flight per segment: the output, the saved activations and the gradient buffer are static, so a second forward call before
    the first one's backward would overwrite what that backward reads (VALOR calls every encoder once per forward pass; accumulation
    micro-steps run forward + backward one after the other)."""

    def __init__(self, name, fn, warmup=2, max_keys=None):
        import os
        self.name, self.fn, self.warmup = name, fn, warmup
        # every captured key keeps a full set of saved activations alive: a job that meets many shapes (multi-task batches, a partial
        # last batch, drop-path on / off) must not multiply the encoder's activation memory without bound. Keys beyond the cap run eagerly.
        self.max_keys = int(os.environ.get("VALOR_GRAPH_MAX_KEYS", "4")) if max_keys is None else int(max_keys)
        self.anchor = None
        self.pool = None
        self.calls = {}
        self.captured = {}

    def __call__(self, *xs, key_extra=None):
        assert ops.DropoutState.base is not None, "graph capture needs ops.DropoutState's device mode (VALOR.enable_graphs)"
        key = tuple((tuple(x.shape), x.dtype, bool(x.requires_grad)) for x in xs) + (ops.DropoutState.offset, key_extra)
        if self.anchor is None:
            self.anchor = torch.zeros(1, device=xs[0].device, requires_grad=True)
        cap = self.captured.get(key)
        if cap is None:
            if len(self.captured) >= self.max_keys:
                return self.fn(*xs)
            if len(self.calls) > 64:               # keys that never repeat (a driver that does not restart the by-value offsets): do not grow
                self.calls.clear()
            n = self.calls.get(key, 0)
            self.calls[key] = n + 1
            if n < self.warmup:
                return self.fn(*xs)
            off0 = ops.DropoutState.offset
            try:
                cap = self._capture(xs)
            except torch.cuda.OutOfMemoryError:
                # no room for a private pool of this encoder's activations: this segment stays eager from here on
                ops.GradSink.recorder = None
                ops.DropoutState.offset = off0
                self.max_keys = len(self.captured)
                torch.cuda.empty_cache()
                return self.fn(*xs)
            self.captured[key] = cap
            self.calls.pop(key, None)
        return _Replay.apply(cap, self.anchor, *xs)

    def _capture(self, xs):
        import os
        if os.environ.get("VALOR_GRAPH_DEBUG"):
            print(f"[valor_amd.graphs] capturing {self.name}: call counts {dict((str(k)[-60:], v) for k, v in self.calls.items())}", flush=True)
        cap = _Captured()
        ctx = _capture_ctx(xs[0].device)
        cap.stream = ctx["stream"]
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        from . import kernels as K
        with torch.cuda.stream(cap.stream):        # the capture stream's kernel scratch exists BEFORE the capture: it must not come out of
            K.workspace(xs[0].device)              # the graph's private pool (256 MiB + 1 GiB per segment)
            K.ReduceQueue.current(xs[0].device)
        cap.static_in = [x.detach().clone().requires_grad_(bool(x.requires_grad)) for x in xs]
        cap.offset0 = ops.DropoutState.offset
        cap.g_fwd, cap.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # capture_error_mode: "thread_local" when a process group is up -- ProcessGroupNCCL's watchdog thread queries events while we
        # capture, which the default "global" mode treats as an error of the capture (the documented DDP + graphs caveat); the capture
        # itself is unaffected (work launched into a capturing stream is captured from whichever thread launches it: autograd's worker)
        mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
        with torch.cuda.graph(cap.g_fwd, pool=self.pool, stream=cap.stream, capture_error_mode=mode):
            out = self.fn(*cap.static_in)
        cap.draws = ops.DropoutState.offset - cap.offset0
        ops.DropoutState.offset = cap.offset0                  # the replay of this very call advances it again
        cap.static_out = out
        cap.static_gout = torch.zeros_like(out)
        rec = []
        ops.GradSink.recorder = rec
        try:
            with torch.cuda.graph(cap.g_bwd, pool=self.pool, stream=cap.stream, capture_error_mode=mode):
                torch.autograd.backward((out,), (cap.static_gout,))
        finally:
            ops.GradSink.recorder = None
        cap.sunk = rec
        cap.outer_hooked = None
        cap.static_out = out.detach()          # the same memory without the (consumed) capture-time autograd history
        return cap

    def release(self):
        self.captured.clear()
        self.calls.clear()
        self.pool = None
