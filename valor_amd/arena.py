"""Flat parameter / gradient arenas.

All parameters of the model are views into ONE flat buffer of the compute dtype and all gradients
views into ONE flat gradient buffer (sized for 288 GB HBM: nothing is ever re-packed). This gives
  * the fused multi-tensor AdamW a single launch over [master | m | v | grad | param] arenas,
  * the data-parallel reducer contiguous bucket ranges (RCCL all-reduce straight on the arena),
  * zero-copy packed views (fused QKV = adjacent q/k/v tensors).
Every tensor starts at a multiple of the optimizer chunk (1024 elements) so a per-chunk int8 table
maps elements to their optimizer param group.
"""
import torch
from torch import nn

from . import lib


def _chunk():
    return lib.load().valor_adamw_chunk()


class ParamArena:
    def __init__(self, entries, dtype, device):
        """entries: ordered list of (name, shape, group_id). Creates nn.Parameters named `name`."""
        self.chunk = _chunk()
        self.dtype, self.device = dtype, torch.device(device)
        self.offsets, self.groups = {}, {}
        off = 0
        for name, shape, gid in entries:
            n = 1
            for s in shape:
                n *= s
            self.offsets[name] = (off, n, tuple(shape))
            self.groups[name] = gid
            off += (n + self.chunk - 1) // self.chunk * self.chunk
        self.numel = off
        self.flat = torch.zeros(off, dtype=dtype, device=self.device)
        self.grad = torch.zeros(off, dtype=dtype, device=self.device)
        self.params = {}
        for name, (o, n, shape) in self.offsets.items():
            p = nn.Parameter(self.flat[o:o + n].view(shape), requires_grad=True)
            p.grad = self.grad[o:o + n].view(shape)
            p._arena_name = name
            self.params[name] = p

    def chunk_group_table(self, active=None):
        """int8 [numel / chunk]: optimizer group of each chunk, -1 for chunks of inactive tensors."""
        t = torch.full((self.numel // self.chunk,), -1, dtype=torch.int8)
        for name, (o, n, _) in self.offsets.items():
            if active is not None and name not in active:
                continue
            c0, c1 = o // self.chunk, (o + n + self.chunk - 1) // self.chunk
            t[c0:c1] = self.groups[name]
        return t.to(self.device)

    def rebind_grads(self):
        """(re)attach .grad views (autograd accumulates in place into the arena)."""
        for name, (o, n, shape) in self.offsets.items():
            self.params[name].grad = self.grad[o:o + n].view(shape)

    def grads_bound(self):
        """True if every parameter's .grad is still its arena view (a stock zero_grad(set_to_none=True) or `p.grad = None`
        detaches them: autograd would then allocate gradients OUTSIDE the arena and the optimizer / reducer would read zeros)."""
        base, esz = self.grad.data_ptr(), self.grad.element_size()
        for name, (o, n, _) in self.offsets.items():
            g = self.params[name].grad
            if g is None or g.data_ptr() != base + o * esz:
                return False
        return True

    def range_of(self, name):
        o, n, _ = self.offsets[name]
        return o, o + (n + self.chunk - 1) // self.chunk * self.chunk
