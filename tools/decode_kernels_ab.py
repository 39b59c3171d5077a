"""Kernels of the cached decoding step, timed with HIP events inside a captured graph of 20 back-to-back launches (the way the step runs them):
the few-row GEMM (family 5) at the decoder's shapes for 128 (greedy) and 384 (beam-3) rows, and the cached self-attention at 2 query rows.
Switches: VALOR_SKINNY_TILE="mt,ns" (pins the workgroup tile of family 5: 16 mt rows x 16 ns columns), VALOR_ATTN_VARIANT (7: one-wave decode kernel, 3: tiled).
usage: python tools/decode_kernels_ab.py out.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
if "VALOR_ATTN_VARIANT" in os.environ:
    so.valor_attn_set_variant(int(os.environ["VALOR_ATTN_VARIANT"]))
pol = K.infer_policy()
res = {"skinny_tile": os.environ.get("VALOR_SKINNY_TILE", "auto"), "attn_variant": so.valor_attn_set_variant(-1), "gemm_us": {}, "attn_us": {}}


def graph_time(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return round(best, 2)


for M in (128, 384):
    for N, Kd in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (30522, 768)):
        a = torch.randn((M, Kd), device=dev).bfloat16()
        w = torch.randn((N, Kd), device=dev).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        res["gemm_us"][f"M{M}_N{N}_K{Kd}"] = graph_time(lambda: K.gemm(a, w, out=out, policy=pol))
for R in (64, 192):
    for L in (41, 100):
        E = 768
        qkv = torch.randn((R, 2, 3 * E), device=dev).bfloat16()
        c = torch.randn((R, L, 2 * E), device=dev).bfloat16()
        mask = torch.zeros((R, 2, L), device=dev)
        mask[:, :, L // 2:] = -10000.0
        res["attn_us"][f"R{R}_L{L}"] = graph_time(lambda: K.attn_fwd(qkv[:, :, :E], c[:, :, :E], c[:, :, E:], 12, mask=mask, scale=0.125))
print(json.dumps(res))
json.dump(res, open(sys.argv[1], "w"), indent=1)
