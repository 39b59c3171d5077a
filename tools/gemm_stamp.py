"""Per-wave cycle stamps of the family-4 GEMM (diagnostic library built by tools/build_stamp_lib.sh, loaded through VALOR_HIP_LIB; SCHED=0/1 picks the schedule, MFMA32=1 the 32 x 32 x 16 main loop of the NN layout):
where a 256 x 128 tile's life goes -- prologue (launch -> first operands landed), K loop, drain, epilogue -- and how the two workgroups
of a CU overlap. usage: VALOR_HIP_LIB=valor_amd/libvalor_hip_stamp.so python tools/gemm_stamp.py M N K ta tb [out.json]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

M, N, Kd, ta, tb = [int(x) for x in sys.argv[1:6]]
dev = torch.device("cuda:0")
so = lib.load()
so.valor_gemm_set_policy(8, 1)
so.valor_gemm_set_narrow_sched(int(os.environ.get("SCHED", "1")))
so.valor_gemm_set_policy(9, int(os.environ.get("MFMA32", "0")))       # NN layout: main loop on v_mfma_f32_32x32x16_bf16
WIDE = int(os.environ.get("WIDE", "0"))                                 # 1: the eight-wave kernel of csrc/gemm8w.hip (policy key 10)
so.valor_gemm_set_policy(10, WIDE)
NW = 8 if WIDE else 4
A = torch.randn((Kd, M) if ta else (M, Kd), device=dev).bfloat16()
B = (0.05 * torch.randn((Kd, N) if tb else (N, Kd), device=dev)).bfloat16()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
assert so.valor_gemm_kernel_for(0, ta, tb, M, N, Kd, 0) == 4
for _ in range(3):
    K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
e1.record()
torch.cuda.synchronize()
kernel_us = e0.elapsed_time(e1) / 5 * 1e3
ws = K.workspace(dev)
nblk = ((M + 255) // 256) * ((N + 127) // 128)
raw = ws.view(torch.int64)[: nblk * NW * 24].cpu().numpy().reshape(nblk, NW, 24)
st = raw[:, :, :5].astype(np.float64)
hw, xcc = raw[:, :, 5], raw[:, :, 6] & 0xf
seg = {"prologue": st[:, :, 1] - st[:, :, 0], "k_loop": st[:, :, 2] - st[:, :, 1], "drain_sync": st[:, :, 3] - st[:, :, 2],
       "epilogue": st[:, :, 4] - st[:, :, 3], "total": st[:, :, 4] - st[:, :, 0]}
res = {"MNK": [M, N, Kd], "ta": ta, "tb": tb, "blocks": nblk, "k_tiles": Kd // 64, "kernel_us_with_stamps": round(kernel_us, 1),
       "lib": os.environ.get("VALOR_HIP_LIB", "in-tree")}
for k, v in seg.items():
    res[k] = {"median": float(np.median(v)), "p10": float(np.percentile(v, 10)), "p90": float(np.percentile(v, 90))}
res["k_loop_per_tile_median"] = res["k_loop"]["median"] / (Kd // 64)
if Kd // 64 > 5 and not WIDE:
    hs = raw[:, :, 8:19].astype(np.float64)
    names = ["H0", "H1", "wait+barrier(a)", "H2", "H3", "H4", "wait+barrier(b)", "H5", "H6", "H7"]
    dd = np.diff(hs, axis=2)
    res["k_tile_5_segments_median"] = {f"{i}:{n}": float(np.median(dd[:, :, i])) for i, n in enumerate(names)}
    res["k_tile_5_total_median"] = float(np.median(hs[:, :, 10] - hs[:, :, 0]))
t0, t1 = st[:, :, 0].min(), st[:, :, 4].max()
res["kernel_span_cycles"] = float(t1 - t0)
# memtime ticks at a constant 100 MHz on this part? compare with the MFMA count: 64 MFMAs x 16 cycles per K-tile per wave
res["mfma_cycles_per_tile_per_wave"] = 512 if WIDE else 1024
res["waves_per_workgroup"] = NW
# overlap: for each CU (xcc, se, sh, cu from HW_ID) the fraction of the span in which >= 1 workgroup is inside its K loop
cu = ((hw[:, 0] >> 8) & 0xf) | (((hw[:, 0] >> 12) & 0x1) << 4) | (((hw[:, 0] >> 13) & 0x7) << 5) | (xcc[:, 0] << 8)
busy = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    ev = sorted([(st[i, 0, 1], 1) for i in idx] + [(st[i, 0, 2], -1) for i in idx])
    depth, last, acc1, acc2 = 0, None, 0.0, 0.0
    for t, d in ev:
        if last is not None:
            if depth >= 1: acc1 += t - last
            if depth >= 2: acc2 += t - last
        depth += d; last = t
    span = max(st[i, 0, 4] for i in idx) - min(st[i, 0, 0] for i in idx)
    busy.append((acc1 / span, acc2 / span, len(idx)))
busy = np.array(busy)
res["cus_seen"] = int(len(busy))
res["frac_span_with_ge1_wg_in_k_loop"] = float(busy[:, 0].mean())
res["frac_span_with_2_wg_in_k_loop"] = float(busy[:, 1].mean())
res["blocks_per_cu_mean"] = float(busy[:, 2].mean())
if os.environ.get("STAMP_RAW"):
    np.save(os.environ["STAMP_RAW"], raw)
print(json.dumps(res, indent=1))
if len(sys.argv) > 6:
    json.dump(res, open(sys.argv[6], "w"), indent=1)
