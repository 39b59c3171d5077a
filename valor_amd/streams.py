"""A second compute stream for the encoders that do not depend on each other.

The AST audio encoder and the CLIP text tower (M = 16.5 k / 2 k token rows: their GEMM grids fill 3/4 of a round, their small kernels a
fraction of one) are independent of the ViT until the contrastive heads / the decoder. Issued on a side stream they run beside the ViT's
chip-filling kernels and soak up its tail rounds; autograd runs every backward node on the stream of its forward op, so the backward
overlaps the same way (the reference is single stream: model/pretrain.py:246-263 runs the three encoders one after the other).

What makes this safe here: per-stream kernel workspaces (kernels.workspace), the reducer waits for every compute stream before a bucket
leaves (dist.Reducer._launch), tensors that cross streams are recorded on the consumer (`join`), and the gradient-arena writes of the side
stream's wgrad kernels -- invisible to autograd -- are joined into the caller's stream by a callback at the end of backward.
VALOR_ENCODER_STREAMS=0 keeps everything on one stream."""
import os

import torch
from torch.autograd import Function
from torch.autograd.variable import Variable

_SIDE = {}


def enabled():
    return os.environ.get("VALOR_ENCODER_STREAMS", "1") != "0"


def side_stream(device):
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _SIDE.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE[key] = s
    return s


_MAIN = {}


def set_main(device):
    """remember the stream a step is issued from (VALOR._forward_groups calls this at the top of every forward)"""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    _MAIN[key] = torch.cuda.current_stream(device)


def compute_streams(device):
    """every stream gradient writes may be in flight on: the step's main stream, the side stream if it exists, and the current one
    (inside a backward node of the side stream the CURRENT stream is the side stream: the main stream must still be waited for)"""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    out = []
    for s in (_MAIN.get(key), _SIDE.get(key), torch.cuda.current_stream(device)):
        if s is not None and all(s != o for o in out):
            out.append(s)
    return out


class _Join(Function):
    @staticmethod
    def forward(ctx, side, *tensors):
        main = torch.cuda.current_stream()
        main.wait_stream(side)
        for t in tensors:
            t.record_stream(main)          # allocated on the side stream, read from here on by kernels of this one
        ctx.side = side
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        side = ctx.side
        # runs once the whole graph has been executed, on the stream backward() was called from
        Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream().wait_stream(side))
        for g in grads:
            if g is not None:
                g.record_stream(side)      # produced here, consumed by the side stream's backward kernels
        return (None,) + grads


def join(side, *tensors):
    """the current stream waits for `side`; returns the tensors (None entries pass through) ready for use on the current stream"""
    live = [t for t in tensors if t is not None]
    if not live:
        torch.cuda.current_stream().wait_stream(side)
        return tensors
    outs = iter(_Join.apply(side, *live))
    return tuple(next(outs) if t is not None else None for t in tensors)


class _Fork(Function):
    """hand a tensor of the current stream to `side`; in backward the current stream waits for EVERYTHING queued on `side` -- the
    consumers there may have accumulated into one gradient buffer after the first of them returned it to autograd (ops.GradSlot)."""

    @staticmethod
    def forward(ctx, side, t):
        side.wait_stream(torch.cuda.current_stream())
        t.record_stream(side)
        ctx.side = side
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        torch.cuda.current_stream().wait_stream(ctx.side)
        return None, g


def fork(side, t):
    return _Fork.apply(side, t)


class _Await(Function):
    @staticmethod
    def forward(ctx, side, ev, t):
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        t.record_stream(cur)
        ctx.side = side
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            g.record_stream(ctx.side)
        return None, None, g


class LazyTensors:
    """tensors produced one after the other on a side stream, each with the event that marks it complete: indexing makes the
    CURRENT stream wait for that one tensor only (a decoder layer starts as soon as ITS K|V projection is done)"""

    def __init__(self, side, items):
        self.side, self.items = side, items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        t, ev = self.items[i]
        return _Await.apply(self.side, ev, t)
