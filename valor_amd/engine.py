"""One optimisation step of the pretraining loop (train_utils.py::conduct_train inner body, :302-364):
forward -> sum of losses -> backward (+ overlapped gradient all-reduce) -> warmup-linear LR -> global-norm
clip -> fused AdamW. Losses stay on the device (the reference's per-step `.item()` syncs, :309, are gone;
read them when you log)."""
import gc
import sys

import torch

from . import dist as vdist
from .ops import DropoutState
from .optim import FusedAdamW, get_lr_sched


class TrainEngine:
    def __init__(self, model, opts, optimizer=None, manage_gc=True, graphs=None):
        self.model, self.opts = model, opts
        # graphs: replay the encoders as hipGraphs (valor_amd/graphs.py) -- the CLIP ViT or the VideoSwin encoder (stochastic-depth factors
        # as a graph input), the AST encoder and the CLIP text tower; the decoder (data-dependent masked-row counts) and the shared-BERT text
        # pass stay eager. Default ON since round 6 (bit-identical to eager; interleaved A/B on two boxes, profiles/r06_graphs_ab_*.txt: equal
        # mean throughput, a quarter of the run-to-run spread, half the host time per step); VALOR_GRAPHS=0 / opts.graphs=False / graphs=False: eager.
        import os as _os
        if graphs is None:
            # (not with activation checkpointing: those are the configurations that fill HBM, and a captured encoder keeps its saved
            # activations in a private pool beside whatever the eager warm-up steps left in the caching allocator)
            graphs = _os.environ.get("VALOR_GRAPHS", "1") != "0" and bool(getattr(opts, "graphs", True)) and not getattr(model, "checkpointing", False)
        if graphs and model.arena.flat.is_cuda:
            model.enable_graphs()
        # The cyclic garbage collector fires in the middle of a forward pass (thousands of short-lived autograd objects per
        # step) and stalls kernel submission for 20-40 ms while the GPU drains. Collect at step boundaries instead: the
        # young generation every step, everything every 64 steps.
        self.manage_gc = manage_gc
        if manage_gc:
            gc.disable()
        self.optimizer = optimizer or FusedAdamW(model, opts)
        self.world = torch.distributed.get_world_size() if vdist.is_dist() else 1
        self.reducer = vdist.Reducer(model.arena)
        if self.world > 1:
            model.gather_fn = vdist.packed_allgather_with_grads
            # torch DDP broadcasts rank 0's parameters when it wraps the model (train_utils.py:232); the reducer replaces DDP, so the
            # replicas are made identical here: parameters, and the optimizer's fp32 masters / moments (bf16 mode / resumed state)
            torch.distributed.broadcast(model.arena.flat, 0)
            for t in ((self.optimizer.master,) if self.optimizer.separate_master else ()) + (self.optimizer.exp_avg, self.optimizer.exp_avg_sq):
                torch.distributed.broadcast(t, 0)
        self.global_step = 0
        # The host needs ~45 ms to issue a step the GPU runs for ~118 ms: unthrottled it runs many steps ahead, and every tensor that
        # crosses streams (held by its record_stream event until the GPU passes it) then exists once per step in flight -- the caching
        # allocator keeps growing by 172 .. 592 MiB segments for as long as the lead grows (hipMalloc stalls inside steps). Two steps
        # of lead keep the GPU fed; the step waits for the end of the step before the previous one.
        self.max_ahead = int(getattr(opts, "max_steps_ahead", 2))
        self._step_events = []
        # per-step host / device timing for whoever asks (bench.py sets `trace = []`): every optimizer step appends
        # {"wait_ms": host time blocked in the `max_ahead` throttle, "host_ms": host time of the whole call, "head" / "tail": timing events
        # recorded on the step's stream in front of its first and behind its last launch}. tail[n].elapsed_time(head[n+1]) is the time the
        # stream sat idle between two steps because the host had not issued the next one yet; head.elapsed_time(tail) is the step's span
        # on the device. None (the default) records nothing.
        self.trace = None
        # Even with the lead bounded, the working set keeps growing for a few steps after the first one (a block that crossed streams
        # is reusable only once the other stream has passed it, so some tensors exist once per step in flight): 2 + 172 + 592 MiB of
        # hipMalloc inside steps 4-6 of a run, each a device-wide stall. After the second optimizer step the engine therefore
        # allocates and frees `alloc_headroom_mb` of device memory once (one large block + a few small-pool segments): the caching
        # allocator keeps the segments and carves the late growth out of them instead of calling hipMalloc in the middle of a step.
        import os
        self.alloc_headroom_mb = int(os.environ.get("VALOR_ALLOC_HEADROOM_MB", getattr(opts, "alloc_headroom_mb", 1536)))
        self._headroom_done = False
        self.grad_norm = float(getattr(opts, "grad_norm", 5.0))
        self._task = None
        self._micro = 0
        # VALOR_DP_CHECK=1 (debug): every rank must have seen the same task sequence in an accumulation window -- the closing micro-step's
        # bucket order (dist.Reducer) is only the same on every rank if it did; checked with one small all-reduce per optimizer step
        self._check_window = _os.environ.get("VALOR_DP_CHECK", "0") == "1"
        self._window_tasks = []
        if getattr(opts, "dataset_mix_type", "random") not in ("random", "round-robin", "accum"):
            raise NotImplementedError(f"dataset_mix_type={opts.dataset_mix_type}")

    def train_step(self, batch, task, accum_steps=1):
        """One iteration of conduct_train's loop body (train_utils.py:302-364). accum_steps = n > 1 is dataset_mix_type='accum'
        (:311-317,341: loss / n, one optimizer step every n iterations, gradients of the n micro-steps -- possibly of different
        tasks -- summed): the arena simply keeps accumulating, the data-parallel reduction runs once per window."""
        model, opt = self.model, self.optimizer
        if not model.arena.grads_bound():
            raise RuntimeError("parameter .grad tensors were detached from the gradient arena (p.grad = None / zero_grad(set_to_none=True) "
                               "on the raw parameters?): call model.zero_grad() (it re-binds) before the next step")
        accum = accum_steps > 1
        if task != self._task:
            self.reducer.reset_task(task)
            self._task = task
        tracing = self.trace is not None and model.arena.flat.is_cuda
        if tracing:
            import time as _time
            t_in = _time.perf_counter()
        wait_s = 0.0
        if self.max_ahead > 0 and len(self._step_events) >= self.max_ahead:
            if tracing:
                t_w = _time.perf_counter()
                self._step_events.pop(0).synchronize()
                wait_s = _time.perf_counter() - t_w
            else:
                self._step_events.pop(0).synchronize()
        if tracing:
            head = torch.cuda.Event(enable_timing=True)
            head.record()
        model.train()
        DropoutState.begin_step()          # device mode (model.enable_graphs): by-value offsets restart, the device counter advances
        micro = self._micro + 1             # committed only once forward + backward went through: an exception in between must not shift the
        last = (not accum) or micro % accum_steps == 0          # accumulation window's boundary for the next call
        if self.world > 1 and self._check_window:
            self._window_tasks.append(task)
        # accumulation window: the micro-steps before the last only accumulate; the LAST one reduces bucket by bucket from its gradient
        # hooks like an ordinary step (a bucket that is complete in the last micro-step holds the sum of the whole window), so the
        # window's reduction overlaps that backward instead of running behind it
        self.reducer.prepare_backward(defer=accum and not last, closing=accum and last)
        loss_dict = model(batch, task=task, compute_loss=True)
        loss = sum(loss_dict.values())
        (loss / accum_steps if accum else loss).backward()
        self._micro = micro
        if last and self.world > 1 and self._check_window:
            self._assert_same_window()
        if tracing and self.world > 1:
            # what of the gradient exchange is NOT hidden behind backward: the stream reaches r0 when its last backward kernel is done and
            # r1 when the last bucket has arrived
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record()
        active = self.reducer.finish_backward(last=last)
        if tracing and self.world > 1:
            r1.record()
        if not last:
            loss_dict["total_loss"] = loss.detach()
            return loss_dict
        self.global_step += 1
        if getattr(self.opts, "num_train_steps", 0):
            ratio = get_lr_sched(self.global_step, self.opts)                      # train_utils.py:344-347
            for g in opt.param_groups:
                g["lr"] = g["init_lr"] * ratio
        opt.step(active_names=active, max_grad_norm=self.grad_norm, world_size=self.world)   # clip :358-360, step :362
        loss_dict["total_loss"] = loss.detach()
        if self.max_ahead > 0 and model.arena.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=tracing)
            ev.record()
            self._step_events.append(ev)
            if tracing:
                self.trace.append({"wait_ms": wait_s * 1e3, "host_ms": (_time.perf_counter() - t_in) * 1e3, "head": head, "tail": ev,
                                   "reduce": (r0, r1) if self.world > 1 else None})
        if not self._headroom_done and self.global_step >= 2:
            self.reserve_headroom()
        if self.manage_gc:
            gc.collect(0 if self.global_step % 64 else 2)
        return loss_dict

    def _assert_same_window(self):
        import zlib
        h = zlib.crc32("|".join(self._window_tasks).encode()) & 0x7fffffff
        self._window_tasks = []
        dev = self.model.arena.flat.device
        t = torch.tensor([h, -h], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        if int(t[0]) != -int(t[1]):
            raise RuntimeError("data-parallel ranks saw different task sequences inside one accumulation window: the gradient buckets of the "
                               "closing micro-step would be reduced in different orders (valor_amd/dist.py)")

    def reserve_headroom(self, mb=None):
        """grow the caching allocator's pools by `mb` MiB of free segments now (see __init__); returns the bytes reserved. A pure
        optimisation: the request is clamped to a quarter of the device memory that is free right now (a configuration that fills HBM
        -- VALOR-large at 16 frames holds 182-191 GB -- must not be pushed over the edge by head-room it does not need), and an
        out-of-memory answer to any of the allocations skips the reservation instead of ending the step: in data parallel one rank
        dying here while the others continue would hang the next collective."""
        self._headroom_done = True
        mb = self.alloc_headroom_mb if mb is None else mb
        dev = self.model.arena.flat.device
        if mb <= 0 or dev.type != "cuda":
            return 0
        from . import streams
        before = torch.cuda.memory_reserved(dev)
        cs = streams.compute_streams(dev)
        free_mb = torch.cuda.mem_get_info(dev)[0] >> 20
        mb = min(mb, free_mb // (4 * max(len(cs), 1)))
        if mb < 64:
            return 0
        # the caching allocator keeps one free list PER STREAM: the late growth happens on the encoders' side stream as well as on the
        # step's stream (the first version reserved on the current stream only and two 172 MiB segments still appeared)
        for st in cs:
            with torch.cuda.stream(st):
                try:
                    blocks = [torch.empty(mb << 20, dtype=torch.uint8, device=dev)]
                    blocks += [torch.empty(1 << 20, dtype=torch.uint8, device=dev) for _ in range(16)]    # small pool (requests <= 1 MiB): 2 MiB segments
                except torch.cuda.OutOfMemoryError:
                    blocks = None
                    break
                del blocks
        return torch.cuda.memory_reserved(dev) - before

    def close(self):
        """release what the engine holds outside the Python heap: the native reducer's communicator / stream / events (dist.Reducer.close);
        idempotent, also run when the engine is collected"""
        r = getattr(self, "reducer", None)
        if r is not None:
            r.close()

    def __del__(self):
        # not at interpreter shutdown: the process group / HIP runtime may be gone by then and a native crash in the communicator's
        # destructor cannot be caught -- call close() (before dist.destroy_process_group) in a driver that wants the resources back
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass
