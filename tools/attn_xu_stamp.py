"""Per-wave cycle stamps of the fused cross-attention backward (attn_xu_bwd_kernel; diagnostic library of `tools/build_stamp_lib.sh attn`
through VALOR_HIP_LIB) at the decoder's bench geometry: where one 64-key tile's life goes, and the kernel's prologue / tile loop / epilogue.
usage: VALOR_HIP_LIB=valor_amd/libvalor_hip_attstamp.so python tools/attn_xu_stamp.py [out.json]"""
import ctypes
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
b, H, Skv, E = int(os.environ.get("XU_B", "64")), 12, 1834, 768
Sv = 1576
g = torch.Generator().manual_seed(0)
kv = (torch.randn((b, Skv, 2 * E), generator=g) * 0.5).bfloat16().to(dev)
k, v = kv[:, :, :E], kv[:, :, E:]
scale, p = 1.0 / math.sqrt(64), 0.1
segs = []
for i, (G, T) in enumerate(((3, 32), (1, 42))):
    B = G * b
    q = (torch.randn((B, T, E), generator=g) * 0.5).bfloat16().to(dev)
    do = torch.randn((B, T, E), generator=g).bfloat16().to(dev)
    kvr = None
    if G == 3:
        kvr = torch.tensor([[(0, Skv), (0, Sv), (Sv, Skv - Sv)][r // b] for r in range(B)], dtype=torch.int32).to(dev)
    o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr, kv_bmod=b, scale=scale, p_drop=p, seed=3 + i, offset=100 * i)
    segs.append(dict(q=q, o=o, lse=lse, dout=do, dq=torch.empty_like(q), kv_range=kvr, seed=3 + i, offset=100 * i))
dkv = torch.empty_like(kv)
run = lambda: K.cross_attn_bwd_fused(segs, k, v, dkv[:, :, :E], dkv[:, :, E:], H, b, scale=scale, p_drop=p)
for _ in range(3):
    assert run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
so = lib.load()
nwg = b * H
buf = np.zeros((nwg, 4, 16), dtype=np.uint64)
so.valor_attn_xu_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert so.valor_attn_xu_read_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf[:, :, :12].astype(np.float64)
names = ["K/V DMA issue", "vmcnt(0)", "barrier A", "S/dP + softmax + dK/dV MFMAs + dS writes", "barrier B", "dQ MFMAs", "dK/dV stores issue", "barrier C"]
d = np.diff(st[:, :, :9], axis=2)
NT = (Skv + 63) // 64
res = {"geometry": [b, H, Skv], "kernel_us_with_stamps": round(us, 1), "workgroups": nwg, "tiles": NT, "lib": os.environ.get("VALOR_HIP_LIB", "in-tree")}
res["tile_segments_median_ticks"] = {f"{i}:{n}": float(np.median(d[:, :, i])) for i, n in enumerate(names)}
res["tile_segments_p90_ticks"] = {f"{i}:{n}": float(np.percentile(d[:, :, i], 90)) for i, n in enumerate(names)}
res["tile_total_median"] = float(np.median(st[:, :, 8] - st[:, :, 0]))
res["prologue_median"] = float(np.median(st[:, :, 10] - st[:, :, 9]))
res["kernel_life_median"] = float(np.median(st[:, :, 11] - st[:, :, 9]))
res["tile_loop_share"] = round(float(np.median((st[:, :, 11] - st[:, :, 10]) / (st[:, :, 11] - st[:, :, 9]))), 3)
# first / second round of workgroups (768 workgroups on 512 slots): start times relative to the earliest
t0 = st[:, 0, 9].min()
start = (st[:, 0, 9] - t0)
res["starts_ticks_percentiles_10_50_90"] = [float(np.percentile(start, q)) for q in (10, 50, 90)]
late = start > 0.25 * (st[:, 0, 11].max() - t0)
res["workgroups_started_late"] = int(late.sum())
res["life_first_round_median"] = float(np.median((st[:, 0, 11] - st[:, 0, 9])[~late]))
res["life_second_round_median"] = float(np.median((st[:, 0, 11] - st[:, 0, 9])[late])) if late.any() else None
res["tile_total_first_round"] = float(np.median((st[:, :, 8] - st[:, :, 0])[~late]))
res["tile_total_second_round"] = float(np.median((st[:, :, 8] - st[:, :, 0])[late])) if late.any() else None
res["span_ticks"] = float(st[:, :, 11].max() - t0)
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
