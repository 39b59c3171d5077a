"""Asynchronous host -> device staging of the small per-step host tensors (token ids, attention masks, gather
indices, labels, kv ranges).

`tensor.to(device)` from pageable memory is a SYNCHRONOUS copy on ROCm: the host blocks until everything already
queued on the stream has finished, the GPU then idles while the host queues the next stretch of kernels. The
reference has the same stall (TokenMasker round trip, modeling.py:171-172). Here every host tensor is copied into a
slot of a pinned ring buffer and sent with a truly asynchronous copy on the current stream; a slot is reused only after
the copies of the step that used it have executed (event per ring segment)."""
import torch


class HostStage:
    def __init__(self, device, segment_bytes=8 << 20, segments=4):
        self.device = torch.device(device)
        self.enabled = self.device.type == "cuda"
        self.seg_bytes, self.nseg = segment_bytes, segments
        self.bufs, self.events = None, None
        self.cur, self.off = 0, 0

    def _init(self):
        self.bufs = [torch.empty(self.seg_bytes, dtype=torch.uint8).pin_memory() for _ in range(self.nseg)]
        self.events = [None] * self.nseg

    def begin_step(self):
        """call once per forward: moves to the next ring segment (waiting, if ever necessary, for its old copies)."""
        if not self.enabled:
            return
        if self.bufs is None:
            self._init()
        if self.off:
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.cur] = ev
        self.cur = (self.cur + 1) % self.nseg
        self.off = 0
        if self.events[self.cur] is not None:
            self.events[self.cur].synchronize()
            self.events[self.cur] = None

    def put(self, t, dtype=None):
        """host tensor -> device tensor (async). Device tensors pass through."""
        if t.is_cuda or not self.enabled:
            t = t.to(self.device)
            return t.to(dtype) if dtype is not None else t
        if dtype is not None:
            t = t.to(dtype)
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if self.bufs is None:
            self._init()
        start = (self.off + 255) & ~255
        if nbytes == 0 or start + nbytes > self.seg_bytes:
            return t.to(self.device)                      # oversized: plain (synchronous) copy
        slot = self.bufs[self.cur][start:start + nbytes].view(t.dtype).view(t.shape)
        slot.copy_(t)
        self.off = start + nbytes
        return slot.to(self.device, non_blocking=True)
