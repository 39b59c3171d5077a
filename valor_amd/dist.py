"""Data-parallel pieces: one process per GPU, torch.distributed over RCCL/xGMI ('nccl' backend on ROCm;
'gloo' on CPU for tests). Replaces utils/distributed.py + torch DDP (train_utils.py:232).

  * packed_allgather_with_grads: ONE all_gather of a packed [feat_t | feat_v | feat_a | tokens] buffer
    instead of the reference's 8 collectives + 4 host syncs per step (ddp_allgather_with_grads ->
    size all_gather + .item() + padded all_gather, utils/distributed.py:38-93, x4). Training uses
    drop_last=True (train_utils.py:591) so per-rank sizes are equal. Backward returns the LOCAL slice of
    the incoming gradient and performs no collective -- exactly utils/distributed.py:62-72.
  * Reducer: gradient all-reduce (SUM; the 1/world mean is folded into the optimizer's gradient scale)
    straight on contiguous ranges of the flat gradient arena, bucketed in reverse execution order and
    launched from post-accumulate-grad hooks on a side stream so it overlaps the rest of backward.
    Parameters that receive no gradient for the current task (find_unused_parameters semantics) are
    learnt on the first (synchronous) step; their arena ranges stay zero and ride along in the buckets.
"""
import torch
import torch.distributed as dist
from torch.autograd import Function


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _PackedGather(Function):
    """ONE collective for all features. Packing is PER SAMPLE -- rank r contributes [b, P] rows, P = the flattened sizes of one sample
    of every feature -- so the gathered [world * b, P] buffer holds each feature as a row-strided VIEW ([world * b, ...] with row stride
    P): nothing is copied out of the receive buffer (the per-rank packing of round 3 needed one strided copy per feature on the critical
    path of the contrastive block). Consumers that need dense rows call .contiguous() themselves."""

    @staticmethod
    def forward(ctx, *feats):
        world, rank = dist.get_world_size(), dist.get_rank()
        b = feats[0].shape[0]
        # per-sample packing: every feature contributes rows of ONE sample; a tensor with another leading dimension (several captions per
        # clip) would still reshape whenever its numel divides by b and then slice the wrong rows in backward
        assert all(f.shape[0] == b for f in feats), [tuple(f.shape) for f in feats]
        ctx.meta = [(f.shape, f[0].numel()) for f in feats]
        ctx.rank, ctx.world = rank, world
        flat = torch.cat([f.reshape(b, -1) for f in feats], dim=1)               # [b, P]: the one packing copy
        P = flat.shape[1]
        out = torch.empty((world * b, P), dtype=flat.dtype, device=flat.device)
        if flat.is_cuda:
            dist.all_gather_into_tensor(out.view(-1), flat.view(-1))
        else:
            dist.all_gather(list(out.view(-1).chunk(world)), flat.view(-1))
        res, o = [], 0
        for shape, n in ctx.meta:
            res.append(out[:, o:o + n].unflatten(1, tuple(shape[1:])) if len(shape) > 1 else out[:, o])
            o += n
        return tuple(res)

    @staticmethod
    def backward(ctx, *grads):
        outs = []
        for g, (shape, _) in zip(grads, ctx.meta):
            b = shape[0]
            outs.append(g[ctx.rank * b:(ctx.rank + 1) * b] if g is not None else None)   # local slice, no collective
        return tuple(outs)


def packed_allgather_with_grads(feat_t, feat_v, feat_a, tokens):
    """Gather contrastive features (with local-slice backward) and the text tokens across ranks.
    `tokens` may live on the host or the device; the gathered tokens are returned ON THE FEATURES' DEVICE (the caller only
    needs `tokens != 0`, so there is no device -> host round trip in the middle of the forward pass)."""
    feats = [f for f in (feat_t, feat_v, feat_a) if f is not None]
    dev, dt = feats[0].device, feats[0].dtype
    # tokens ride in the same buffer as RAW BYTES: int32 ids reinterpreted as elements of the features' dtype (2 bf16 / 1 fp32 per id).
    # A gather only moves bytes (no arithmetic touches them on the way), so any bit pattern survives; they are reinterpreted back below.
    tok = tokens.to(dev).to(torch.int32).contiguous()
    tok_parts = tok.view(dt)
    gathered = _PackedGather.apply(*feats, tok_parts)
    gi = iter(gathered[:-1])
    ft = next(gi) if feat_t is not None else None
    fv = next(gi) if feat_v is not None else None
    fa = next(gi) if feat_a is not None else None
    tokens_all = gathered[-1].detach().contiguous().view(torch.int32).long()       # the gathered feature is a row-strided view: copy first
    return ft, fv, fa, tokens_all


def ddp_allgather(x):
    """utils/distributed.py:77-93 (no gradient, UNEQUAL leading sizes: the last validation batches differ between ranks; test.py:275-290
    gathers features and tokens with it): sizes first, every rank padded to the largest, one gather, the padding cut away again. ONE size
    exchange + ONE payload collective; the result keeps the ranks' order. A single rank returns its input."""
    if not is_dist():
        return x
    world = dist.get_world_size()
    x = x.contiguous()
    size = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    top = max(sizes)
    if x.shape[0] < top:
        x = torch.cat((x, x.new_zeros((top - x.shape[0],) + tuple(x.shape[1:]))), dim=0)
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def all_gather_list(obj):
    """utils/distributed.py:133-170 (pickled python objects of every rank, in rank order): ids, hit counters"""
    if not is_dist():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


class Reducer:
    """mode (env VALOR_REDUCE or the argument):
         "allreduce"  one SUM all-reduce per bucket in the arena's dtype (default; RCCL picks ring / direct per message size);
         "rs_ag"      reduce-scatter + all-gather per bucket (the two halves of a direct all-reduce issued explicitly: every one of a
                      GPU's 7 xGMI links carries 1/world of the bucket in each half) -- nccl / RCCL only (gloo has no
                      reduce_scatter);
         "fp32"       the bucket is widened to fp32 for the cross-rank sum and rounded ONCE on the way back (world - 1 fewer bf16
                      roundings per element at twice the bytes on the wire).
       Default: "allreduce". SURVEY 8(e) argues for direct reduce-scatter + all-gather on the fully connected xGMI mesh (7 links x ~153
       GB/s busy: ~1.2 ms for 0.75 GB against ~8.6 ms for a single-link-bound ring). RCCL chooses its algorithm per message size and
       topology by itself; which one it picks for a 48 MiB bucket on an 8-GPU MI355X node has NOT been measured here (no multi-GPU node
       was available to the builder), so the default is the call that leaves the choice to RCCL and is ONE collective per bucket.
       "rs_ag" forces the two halves of a direct all-reduce (in-place shards) and stays opt-in until a multi-GPU run has timed both:
       VALOR_REDUCE=rs_ag python bench.py --gpus 8 is the A/B; bench.py prints the mode in its line.
       Summing in bf16 across 8 ranks perturbs each element by ~0.4 % rms (like one more bf16 rounding of the gradient) and the global
       norm by < 1e-5 relative: tests/test_dist_cpu.py::test_bf16_cross_rank_sum_error quantifies it."""

    def __init__(self, arena, bucket_bytes=48 << 20, mode=None, native=None):
        import os
        self.mode = mode or os.environ.get("VALOR_REDUCE", "allreduce")
        assert self.mode in ("allreduce", "rs_ag", "fp32"), self.mode
        self.arena = arena
        self.world = dist.get_world_size() if is_dist() else 1
        esz = arena.flat.element_size()
        # buckets = contiguous arena ranges, walked from the END (gradients become ready in reverse order)
        names = list(arena.offsets)
        self.buckets, cur, cur_bytes = [], [], 0
        for n in reversed(names):
            s, e = arena.range_of(n)
            cur.append(n); cur_bytes += (e - s) * esz
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur); cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_range = [(min(arena.range_of(n)[0] for n in b), max(arena.range_of(n)[1] for n in b)) for b in self.buckets]
        self.bucket_of = {n: i for i, b in enumerate(self.buckets) for n in b}
        self.expected = None           # per-bucket set of names that get a grad for the current task
        self.uses = None               # name -> number of gradient writes per backward (learnt on the first step)
        self._tasks = {}               # task key -> (expected, uses): multi-task mixes alternate between a handful of task strings
        self._task = None
        self.window = None             # gradient accumulation: union of the names touched since the last reduction
        self.touched = {}              # name -> writes seen in the current backward
        self.pending, self.works = None, []
        self._launched = set()         # buckets launched since prepare_backward
        self.comm_stream = torch.cuda.Stream() if arena.flat.is_cuda else None
        # native=True / VALOR_REDUCER_NATIVE=1: the collectives are issued by the library's own reducer (csrc/reducer.hip: its RCCL
        # communicator, communication stream and per-bucket events behind valor_reducer_*) instead of torch.distributed work objects.
        # Same buckets, same order, same in-place sums; this class keeps deciding when a bucket is complete. Opt-in: no multi-GPU node
        # was available to the builder, the path has run with one rank on a GPU (tests/test_native_reducer_gpu.py) and nothing more.
        self.native = None
        want = native if native is not None else os.environ.get("VALOR_REDUCER_NATIVE", "0") == "1"
        if want and arena.flat.is_cuda and dist.is_available() and dist.is_initialized() and self.mode in ("allreduce", "rs_ag"):
            self._create_native()
        elif want:
            import warnings
            why = ("mode 'fp32' widens every bucket through a temporary, which the native reducer does not do" if self.mode == "fp32" else
                   "it needs a device arena and an initialised torch.distributed process group")
            warnings.warn(f"native reducer requested (native=True / VALOR_REDUCER_NATIVE=1) but not used: {why}; "
                          "the torch.distributed path runs instead")
        for name, p in arena.params.items():
            if p.requires_grad:                    # frozen parameters (VALOR.frozen_vision / frozen_multimodal) never get a gradient
                p.register_post_accumulate_grad_hook(self._make_hook(name))
        from . import ops
        ops.GradSink.listener = self._on_grad      # kernels that accumulate straight into the arena report here

    def _create_native(self):
        import ctypes
        from . import lib
        g = self.arena.grad
        idt = torch.zeros(128, dtype=torch.uint8, device=g.device)
        if dist.get_rank() == 0:
            host = (ctypes.c_char * 128)()
            lib.call("valor_reducer_unique_id", ctypes.cast(host, ctypes.c_void_p))
            idt.copy_(torch.frombuffer(bytearray(host.raw), dtype=torch.uint8))
        if dist.get_world_size() > 1:
            dist.broadcast(idt, 0)
        idb = (ctypes.c_char * 128).from_buffer_copy(bytes(idt.cpu().numpy().tobytes()))
        nb = len(self.bucket_range)
        offs = (ctypes.c_int64 * nb)(*[s for s, _ in self.bucket_range])
        cnts = (ctypes.c_int64 * nb)(*[e - s for s, e in self.bucket_range])
        handle = ctypes.c_void_p()
        torch.cuda.synchronize(g.device)          # ncclCommInitRank is a collective rendezvous: nothing of ours in flight around it
        lib.call("valor_reducer_create", ctypes.cast(ctypes.byref(handle), ctypes.c_void_p), ctypes.cast(idb, ctypes.c_void_p), dist.get_rank(),
                 dist.get_world_size(), 0 if g.dtype == torch.bfloat16 else 1, g.data_ptr(), ctypes.cast(offs, ctypes.c_void_p),
                 ctypes.cast(cnts, ctypes.c_void_p), nb, 1 if self.mode == "rs_ag" else 0)
        self.native = handle

    def close(self):
        """free the native reducer (communicator, stream, events); the torch.distributed path holds nothing to free"""
        if self.native is not None:
            from . import lib
            lib.call("valor_reducer_destroy", self.native)
            self.native = None

    def _make_hook(self, name):
        def hook(param):
            from . import ops
            if ops.GradSink.recorder is not None:
                # a backward is being captured into a graph (valor_amd/graphs.py). Autograd runs a leaf's post-accumulate hooks ONCE per
                # backward pass the leaf takes part in -- also when every gradient of it was written straight into the arena -- so a
                # parameter used inside the captured segment AND by eager code of the same step (the shared-BERT text pass beside the
                # graphed decoder stack) is hooked once in eager mode but once per (captured backward, outer backward) otherwise. The
                # replay therefore reports a hook-origin entry only for names the outer backward does not hook itself.
                ops.GradSink.recorder.append(("hook", name))
                return
            ops.GradSink.live_hooks.add(name)
            self._on_grad(name)
        return hook

    def _on_grad(self, name):
        """one gradient contribution to `name` has been enqueued (autograd AccumulateGrad or a direct arena write)."""
        from . import ops
        if ops.GradSink.recorder is not None:      # a backward is being captured into a graph: nothing ran, the replay reports it
            ops.GradSink.recorder.append(name)
            return
        n = self.touched.get(name, 0) + 1
        self.touched[name] = n
        if self.pending is None or n < self.uses.get(name, 1):
            return
        i = self.bucket_of[name]
        s = self.pending[i]
        s.discard(name)
        if not s and self.expected[i]:
            self._launch(i)

    def _reduce(self, buf):
        """SUM `buf` (a contiguous range of the gradient arena) over the ranks, in place; returns the async works to wait for"""
        if self.mode == "fp32" and buf.dtype != torch.float32:
            wide = buf.float()
            dist.all_reduce(wide, op=dist.ReduceOp.SUM)
            buf.copy_(wide)
            return []
        if self.mode == "rs_ag" and buf.numel() % self.world == 0:       # arena ranges are multiples of 1024 elements
            shard = buf.view(self.world, -1)[dist.get_rank()]
            dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM)       # in place: the output is this rank's slice of the input
            return [dist.all_gather_into_tensor(buf, shard, async_op=True)]
        return [dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)]

    def _launch(self, i):
        self._launched.add(i)
        if self.world == 1 and self.native is None:
            return
        s, e = self.bucket_range[i]
        buf = self.arena.grad[s:e]
        if self.native is not None:
            import ctypes
            from . import lib, streams
            cs = streams.compute_streams(buf.device)
            arr = (ctypes.c_void_p * len(cs))(*[c.cuda_stream for c in cs])
            lib.call("valor_reducer_launch_bucket", self.native, i, ctypes.cast(arr, ctypes.c_void_p), len(cs))
            self._native_pending = True
            return
        if self.comm_stream is not None:
            from . import streams
            evs = []
            for cs in streams.compute_streams(buf.device):      # gradient writes may be in flight on the encoders' side stream too
                ev = torch.cuda.Event(); ev.record(cs); evs.append(ev)
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                self.works += self._reduce(buf)
        else:
            self.works += self._reduce(buf)

    def _wait_all(self):
        """the current stream waits for every bucket in flight"""
        if self.native is not None:
            if getattr(self, "_native_pending", False):
                from . import lib
                lib.call("valor_reducer_wait", self.native, torch.cuda.current_stream().cuda_stream)
                self._native_pending = False
            return
        for w in self.works:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def prepare_backward(self, defer=False, closing=False):
        from . import ops as _ops
        _ops.GradSink.live_hooks.clear()
        """defer=True: a micro-step of a gradient accumulation window -- gradients keep accumulating in the arena and NO bucket
        is reduced now (reducing a partially accumulated arena twice would count the earlier micro-steps world times).
        closing=True: the LAST micro-step of a window (train_utils.py:311-329: the reference's DDP reduces in the backward that precedes
        the optimizer step). It runs like an ordinary step -- a bucket that completes in this backward holds the sum of the whole
        window and is launched from the hooks, overlapped with the rest of the backward -- and finish_backward() sends whatever the
        window touched that this micro-step did not (another task's parameters) behind it. Without a closing micro-step
        (finish_backward(last=True) on a deferred one) the window is reduced at once, un-overlapped."""
        self.touched = {}
        self.works = []
        self._launched = set()
        self.defer = defer
        self.closing = closing and not defer
        self.pending = [set(x) for x in self.expected] if (self.expected is not None and not defer) else None

    def finish_backward(self, last=True):
        """Wait for the bucket all-reduces (or, on the first step of a task, reduce everything at once and
        learn which parameters are used and how many gradient writes each receives).
        Accumulation windows (prepare_backward(defer=True)): micro-steps only record what they touched; the last one
        (`last=True`) reduces the whole arena once and returns the union of the touched names."""
        if getattr(self, "defer", False):
            self.window = (self.window or set()) | set(self.touched)
            self.pending = None
            if not last:
                return set()
            if self.native is not None:              # the window's one reduction, bucket by bucket on the native reducer
                for i in range(len(self.buckets)):
                    self._launch(i)
                self._wait_all()
            elif self.world > 1:
                for w in self._reduce(self.arena.grad):
                    w.wait()
            names, self.window = self.window, None
            return names
        if self.expected is None or self.touched != self.uses:
            if self.pending is not None:       # the used-parameter set changed (new task): redo synchronously
                for w in self.works:
                    w.wait()
                diff = {n: (self.touched.get(n, 0), self.uses.get(n, 0)) for n in set(self.touched) | set(self.uses)
                        if self.touched.get(n, 0) != self.uses.get(n, 0)}
                raise RuntimeError("used-parameter set changed between steps; call reset_task() when switching tasks "
                                   f"[{len(diff)} names differ, (writes this step, writes learnt): {dict(list(diff.items())[:8])}]")
            self.uses = dict(self.touched)
            self.expected = [set(n for n in b if n in self.touched) for b in self.buckets]
            if self.world > 1:
                # first step of a task: backward is over, nothing to overlap with -- but the buckets still go out one by one,
                # asynchronously on the communication stream (the same launches every later step makes from the hooks), instead of
                # one synchronous whole-arena collective: the bucket pipeline of RCCL is warm for step 2 and the host does not block
                # on 0.75 GB before it queues the optimizer
                self.works = []
                for i in range(len(self.buckets)):
                    self._launch(i)
                self._wait_all()
        elif not getattr(self, "closing", False):
            self._wait_all()
        self.pending = None
        if getattr(self, "closing", False):
            # the window's last micro-step: buckets that hold names of EARLIER micro-steps only (a window that mixes tasks) were not
            # launched by this backward's hooks; they go out now, then everything is waited for
            names = (self.window or set()) | set(self.touched)
            self.window, self.closing = None, False
            if self.world > 1 or self.native is not None:
                for i in sorted({self.bucket_of[n] for n in names} - self._launched):
                    self._launch(i)
            self._wait_all()
            return names
        return set(self.touched)

    def reset_task(self, task=None):
        """switch to another task string: its used-parameter set is restored if this reducer has seen the task before (so the
        overlapped path resumes at once instead of re-learning with a synchronous whole-arena reduction on every switch)"""
        if self._task is not None and self.expected is not None:
            self._tasks[self._task] = (self.expected, self.uses)
        self._task = task
        self.expected, self.uses = self._tasks.get(task, (None, None)) if task is not None else (None, None)
