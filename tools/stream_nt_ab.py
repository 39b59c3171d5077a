"""A/B of non-temporal accesses in the streaming (HBM-bound) kernels: fused LayerNorm forward / backward (valor_ln_set_nt bit mask) and
the fused AdamW update (valor_adamw_set_nt), at the VALOR-base sizes, interleaved rounds, HIP events. usage: python tools/stream_nt_ab.py out.json"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402
from valor_amd.kernels import _ptr, _stream  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
res = {}


def timeit(fn, n=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, cols, p in [(100864, 768, 0.0), (100864, 768, 0.1), (16512, 768, 0.1)]:
    x = torch.randn((rows, cols), device=dev).bfloat16(); r = torch.randn_like(x); bias = torch.randn(cols, device=dev).bfloat16()
    g = torch.randn(cols, device=dev).bfloat16(); be = torch.randn(cols, device=dev).bfloat16()
    dy = torch.randn_like(x); dz = torch.randn_like(x)
    z, y, mean, rstd = K.bdrln_fwd(x, bias, r, g, be, 1e-5, p_drop=p, seed=1, offset=0)
    nb = rows * cols * 2
    fw, bw = {}, {}
    for rnd in range(3):
        for m in (0, 1, 2, 4, 3, 6, 7):
            so.valor_ln_set_nt(m)
            fw.setdefault(m, []).append(timeit(lambda: K.bdrln_fwd(x, bias, r, g, be, 1e-5, p_drop=p, seed=1, offset=0)))
        for m in (0, 8, 16, 24):
            so.valor_ln_set_nt(m)
            bw.setdefault(m, []).append(timeit(lambda: K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=0, want_dbias=True)))
    so.valor_ln_set_nt(0)
    key = f"ln_{rows}x{cols}_p{p}"
    res[key] = {"fwd_us_by_mask": {m: round(sorted(v)[1], 1) for m, v in fw.items()}, "bwd_us_by_mask": {m: round(sorted(v)[1], 1) for m, v in bw.items()},
                "fwd_TBps_mask0": round(4 * nb / sorted(fw[0])[1] / 1e6, 2)}
    print(key, res[key], flush=True)

# AdamW over 374.7 M parameters (VALOR-base): 28 B per parameter
n = 374_784_000 // 1024 * 1024
master = torch.randn(n, device=dev); m1 = torch.zeros(n, device=dev); v1 = torch.zeros(n, device=dev)
grad = (0.01 * torch.randn(n, device=dev)).bfloat16(); param = master.bfloat16()
table = torch.zeros(n // 1024, dtype=torch.int8, device=dev)
gscale = torch.ones((), device=dev)
lr = (ctypes.c_float * 10)(*([1e-4] * 10)); wd = (ctypes.c_float * 10)(*([0.01] * 10))
step = lambda: lib.call("valor_adamw", _stream(), 0, _ptr(master), _ptr(m1), _ptr(v1), _ptr(grad), _ptr(param), _ptr(table), n, lr, wd, 10,
                        0.9, 0.98, 1e-6, 3, 1, _ptr(gscale), 1)
ad = {}
for rnd in range(3):
    for m in (0, 1, 2, 3):
        so.valor_adamw_set_nt(m)
        ad.setdefault(m, []).append(timeit(step, 5))
so.valor_adamw_set_nt(0)
res["adamw_374.8M"] = {"us_by_mask": {m: round(sorted(v)[1], 1) for m, v in ad.items()}, "TBps_mask0": round(28.0 * n / sorted(ad[0])[1] / 1e6, 2)}
print("adamw", res["adamw_374.8M"], flush=True)
json.dump(res, open(sys.argv[1], "w"), indent=1)
