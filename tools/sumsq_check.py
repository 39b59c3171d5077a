"""valor_grad_norm_clip at the VALOR-base arena size (342 M bf16 gradients, some chunks inactive): total norm against torch, HIP-event time.
usage: python tools/sumsq_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import lib  # noqa: E402
from valor_amd.kernels import _ptr, _stream  # noqa: E402

dev = torch.device("cuda:0")
n = 334_000 * 1024
g = (torch.randn(n, device=dev) * 0.01).to(torch.bfloat16)
table = torch.zeros(n // 1024, dtype=torch.int8, device=dev)
table[1000:1500] = -1
table[-3:] = -1
g[1000 * 1024:1500 * 1024] = float("nan")          # inactive chunks may hold anything
partial = torch.empty(4096, dtype=torch.float32, device=dev)
tn = torch.empty(1, dtype=torch.float32, device=dev)
gs = torch.empty(1, dtype=torch.float32, device=dev)
run = lambda: lib.call("valor_grad_norm_clip", _stream(), 0, _ptr(g), _ptr(table), n, 1.0, 5.0, _ptr(partial), _ptr(tn), _ptr(gs))
run()
mask = (table >= 0).repeat_interleave(1024)
ref = torch.where(mask, g.float(), torch.zeros((), device=dev)).double().pow(2).sum().sqrt().item()
print("total_norm", tn.item(), "reference", ref, "rel err", abs(tn.item() - ref) / ref)
assert abs(tn.item() - ref) / ref < 1e-5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"norm + finalize: {us:.1f} us = {n * 2 / us / 1e6:.2f} TB/s")
