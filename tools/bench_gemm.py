"""Micro-benchmark of valor_gemm on the GEMM shapes of the VALOR-base step (per-GPU batch 64).
Writes TF/s per shape; HIP-event timed. Usage: python tools/bench_gemm.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from valor_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
b = 64
SHAPES = [
    # name, M, N, K, ta, tb
    ("vit_qkv_fwd", b * 8 * 197, 2304, 768, 0, 0),
    ("vit_fc1_fwd", b * 8 * 197, 3072, 768, 0, 0),
    ("vit_fc2_fwd", b * 8 * 197, 768, 3072, 0, 0),
    ("vit_proj_fwd", b * 8 * 197, 768, 768, 0, 0),
    ("vit_fc1_dgrad", b * 8 * 197, 768, 3072, 0, 1),
    ("vit_fc2_dgrad", b * 8 * 197, 3072, 768, 0, 1),
    ("vit_fc1_wgrad", 3072, 768, b * 8 * 197, 1, 1),
    ("vit_qkv_wgrad", 2304, 768, b * 8 * 197, 1, 1),
    ("vit_proj_wgrad", 768, 768, b * 8 * 197, 1, 1),
    ("xkv_fwd", b * 1834, 1536, 768, 0, 0),
    ("dec_fc1_fwd", 3 * b * 32, 3072, 768, 0, 0),
    ("head_decoder", 2400, 30522, 768, 0, 0),
]


def run(dtype):
    res = {}
    for name, M, N, Kd, ta, tb in SHAPES:
        if dtype == torch.float32 and M * Kd > 2e8:
            continue
        A = torch.randn((Kd, M) if ta else (M, Kd), device=dev).to(dtype)
        B = torch.randn((Kd, N) if tb else (N, Kd), device=dev).to(dtype)
        out = torch.empty((M, N), dtype=dtype, device=dev)
        for _ in range(2):
            K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        tf = 2.0 * M * N * Kd / ms / 1e9
        res[name] = {"ms": round(ms, 4), "TF": round(tf, 1)}
        print(f"{str(dtype):16s} {name:16s} M={M:7d} N={N:6d} K={Kd:7d}  {ms:8.3f} ms  {tf:8.1f} TF/s", flush=True)
        del A, B, out
    return res


if __name__ == "__main__":
    out = {"bf16": run(torch.bfloat16), "f32": run(torch.float32)}
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
