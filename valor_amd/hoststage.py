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


class PrefetchLoader:
    """Overlap the host -> device copy of the NEXT batch's pixels / spectrograms with the current step (data/loader.py:154-212,
    the apex-derived PrefetchLoader of the reference): the big float tensors of a batch (`video_pixels`, `audio_spectrograms`)
    are copied into pinned double buffers and sent on a side stream while the model works on the previous batch; the token
    tensors stay on the host (the maskers and mask builders read them there, modeling.py:134-174 / valor.py) and everything
    else passes through. The consumer's stream waits for the copy right before the batch is handed over, and the device
    tensors are recorded on it (the caching allocator may otherwise recycle them while a kernel still reads them).
    (name, batch) pairs of the reference's MetaLoader are supported as well as bare batch dicts."""
    DEVICE_KEYS = ("video_pixels", "audio_spectrograms")

    def __init__(self, loader, device="cuda"):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = [{}, {}]
        self._slot_events = [None, None]      # "the copies out of this pinned slot have finished"
        self._flip = 0

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def _stage(self, item):
        name, batch = item if isinstance(item, tuple) else (None, item)
        out = dict(batch)
        slot = self._flip
        pins = self._pinned[slot]
        self._flip ^= 1
        # the H2D copies of the batch staged two calls ago read this pinned slot asynchronously: the host must not overwrite it before
        # they have finished (the GPU-side wait_stream in __iter__ orders kernels, not the host's writes)
        ev = self._slot_events[slot]
        if ev is not None:
            ev.synchronize()
        with torch.cuda.stream(self.stream):
            for k in self.DEVICE_KEYS:
                t = batch.get(k)
                if t is None or t.is_cuda:
                    continue
                buf = pins.get(k)
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    buf = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                    pins[k] = buf
                buf.copy_(t)
                out[k] = buf.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._slot_events[slot] = ev
        return (name, out) if name is not None else out

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            cur = nxt
            batch = cur[1] if isinstance(cur, tuple) else cur
            for k in self.DEVICE_KEYS:
                if isinstance(batch.get(k), torch.Tensor) and batch[k].is_cuda:
                    batch[k].record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._stage(next(it))          # the next batch's copies fly while the caller trains on `cur`
            except StopIteration:
                nxt = None
            yield cur
