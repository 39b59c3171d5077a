"""Per-wave cycle stamps of the persistent self-attention backward (attn_res_bwd_pipe_kernel; diagnostic library built by
`tools/build_stamp_lib.sh attn`, loaded through VALOR_HIP_LIB): where one steady-state (batch, head) item's life goes -- operand loads, barriers,
the dQ loop, the dK / dV loop, stores. usage: VALOR_HIP_LIB=valor_amd/libvalor_hip_attstamp.so python tools/attn_stamp.py [out.json] (B H S from env)"""
import ctypes
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

B, H, S = int(os.environ.get("B", 512)), int(os.environ.get("H", 12)), int(os.environ.get("S", 197))
dev = torch.device("cuda:0")
so = lib.load()
MODE = int(os.environ.get("MODE", "1"))          # 1: the second version of the pipelined kernel, 3: the first
so.valor_attn_set_res_pipeline(MODE)
scale = 1.0 / math.sqrt(64)
g = torch.Generator().manual_seed(S)
E = H * 64
qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
o, lse = K.attn_fwd(q, k, v, H, scale=scale)
dqkv = torch.empty_like(qkv)
run = lambda: K.attn_bwd(q, k, v, o, lse, dout, H, dq=dqkv[:, :, :E], dk=dqkv[:, :, E:2 * E], dv=dqkv[:, :, 2 * E:], scale=scale)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
ncu = torch.cuda.get_device_properties(0).multi_processor_count
buf = np.zeros((ncu, 8, 16), dtype=np.uint64)
so.valor_attn_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = so.valor_attn_read_stamps(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
st = buf[:, :, :12].astype(np.float64)
names = ["operand loads + vmcnt(0)", "barrier 1", "statistics + Q/dO DMA issue", "dQ loop", "dQ stores issue", "vmcnt(0) (Q/dO pieces, dQ stores)", "barrier 2",
         "K/V operand loads + vmcnt(0)", "next K/V DMA issue", "dK/dV loop", "dK/dV stores issue"]
if MODE == 1:
    names = ["phase-1 operands out of the prefetched registers (delta)", "barrier 1", "statistics + Q/dO DMA issue", "dQ loop + K/V fragments out of LDS",
             "vmcnt(0) lgkmcnt(0)", "dQ stores issue", "barrier 2", "-", "next K/V DMA + next phase-1 operand loads issue", "dK/dV loop", "vmcnt(0) + dK/dV stores issue"]
NP = (S + 31) // 32
res = {"mode": MODE, "shape": [B, H, S], "kernel_us_with_stamps": round(us, 1), "items_per_workgroup": B * H / ncu, "us_per_item": round(us / (B * H / ncu), 2),
       "lib": os.environ.get("VALOR_HIP_LIB", "in-tree"), "active_waves": NP}
d = np.diff(st, axis=2)              # [wg][wave][11]
res["segments_median_ticks_active_waves"] = {f"{i}:{n}": float(np.median(d[:, :NP, i])) for i, n in enumerate(names)}
res["segments_by_wave_median"] = {f"wave{w}": [float(np.median(d[:, w, i])) for i in range(11)] for w in range(8)}
res["item_total_median_ticks"] = float(np.median(st[:, :NP, 11] - st[:, :NP, 0]))
# the workgroup's item period: item start of wave 0 to the last wave's stores issued
res["item_span_ticks_median"] = float(np.median(st[:, :NP, 11].max(axis=1) - st[:, :NP, 0].min(axis=1)))
# ticks -> us: the 24 items of a workgroup take the kernel's duration
res["ticks_per_us_if_items_tile_the_kernel"] = round(res["item_span_ticks_median"] / res["us_per_item"], 1)
hw = buf[:, :, 12]
res["simd_of_wave_wg0"] = [int((hw[0, w] >> 4) & 3) for w in range(8)]
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
