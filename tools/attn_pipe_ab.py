"""A/B of the LDS-resident self-attention backward: one workgroup per (batch, head) (mode 0) vs the persistent phase-pipelined kernel (1) vs
one workgroup of 16 waves x 16-row blocks per (batch, head) (2) vs the first version of the pipelined kernel (3)
(valor_attn_set_res_pipeline), at the VALOR-base shapes, interleaved rounds, HIP events. usage: python tools/attn_pipe_ab.py out.json"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
scale = 1.0 / math.sqrt(64)
res = {}
old_mode = so.valor_attn_set_res_pipeline(-1)
for name, B, H, S, p in [("vit_b64 (512 x 12 heads, S = 197)", 512, 12, 197, 0.0), ("ast_b64 (128 x 12 heads, S = 129, dropout 0.1)", 128, 12, 129, 0.1),
                         ("vit_b64 dropout 0.1", 512, 12, 197, 0.1)]:
    g = torch.Generator().manual_seed(S)
    E = H * 64
    qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
    o, lse = K.attn_fwd(q, k, v, H, scale=scale, p_drop=p, seed=5, offset=9)
    dqkv = torch.empty_like(qkv)
    run = lambda: K.attn_bwd(q, k, v, o, lse, dout, H, dq=dqkv[:, :, :E], dk=dqkv[:, :, E:2 * E], dv=dqkv[:, :, 2 * E:], scale=scale, p_drop=p, seed=5, offset=9)
    t = {0: [], 1: [], 2: [], 3: []}
    outs = {}
    for rnd in range(3):
        for mode in (0, 1, 2, 3):
            so.valor_attn_set_res_pipeline(mode)
            run()
            if rnd == 0:
                outs[mode] = dqkv.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / 10 * 1e3)
    so.valor_attn_set_res_pipeline(old_mode)
    fl = 5 * 2.0 * S * S * 64 * B * H
    traffic = 8.0 * B * S * E * 2          # Q K V dO O read, dQ dK dV written
    a, b_, c_ = sorted(t[0])[1], sorted(t[1])[1], sorted(t[2])[1]
    res[name] = {"per_head_us": round(a, 1), "pipelined_us": round(b_, 1), "speedup": round(a / b_, 3),
                 "pipelined_first_version_us": round(sorted(t[3])[1], 1), "second_over_first": round(sorted(t[3])[1] / b_, 3),
                 "versions_bit_identical": bool(torch.equal(outs[1], outs[3])),
                 "sixteen_waves_us": round(c_, 1), "sixteen_waves_over_per_head": round(a / c_, 3), "sixteen_waves_bit_identical": bool(torch.equal(outs[0], outs[2])), "pipelined_TF": round(fl / b_ / 1e6, 1),
                 "pipelined_TBps": round(traffic / b_ / 1e6, 2), "frac_of_mfma_peak": round(fl / b_ / 1e6 / 2500, 3)}
    print(name, res[name], flush=True)
json.dump(res, open(sys.argv[1], "w"), indent=1)
