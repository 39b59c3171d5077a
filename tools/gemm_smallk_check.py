"""8-phase GEMM at the VideoSwin stage-1 / 2 contraction lengths (K = 128 / 256 / 384) under policy keys 7 (forward min K) and 0 (dgrad min K):
kernel family picked, error against fp32 torch.matmul (whole result and worst 256 x 256 tile), time against the 128 x 128 kernels.
usage: python tools/gemm_smallk_check.py out.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from valor_amd import kernels as K, lib  # noqa: E402
from test_gemm_bench_shapes_gpu import _tile_errors  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
res = {}
M = 200704
for name, n, k, tb in (("s1_fc1_fwd NN", 512, 128, False), ("s1_qkv_fwd NN", 384, 128, False), ("s2_fc1_fwd NN", 1024, 256, False), ("s2_qkv_fwd NN", 768, 256, False),
                       ("s1_fc2_dgrad NT", 512, 128, True), ("s2_fc2_dgrad NT", 1024, 256, True), ("s2_qkv_dgrad NT", 256, 768, True)):
    g = torch.Generator().manual_seed(n + k)
    A = torch.randn((M, k), generator=g).to(torch.bfloat16).to(dev)
    B = (0.05 * torch.randn((k, n) if tb else (n, k), generator=g)).to(torch.bfloat16).to(dev)
    bias = None if tb else torch.randn((n,), generator=g).to(torch.bfloat16).to(dev)
    ref = A.float() @ (B.float() if tb else B.float().t()) + (bias.float() if bias is not None else 0.0)
    out = torch.empty((M, n), dtype=torch.bfloat16, device=dev)
    row = {}
    for tag, mink in (("128x128", 4096), ("8-phase", 128)):
        o7, o0 = so.valor_gemm_set_policy(7, mink), so.valor_gemm_set_policy(0, mink)
        try:
            fam = so.valor_gemm_kernel_for(0, 0, int(tb), M, n, k, 0)
            K.gemm(A, B, trans_b=tb, bias=bias, out=out)
            whole, worst = _tile_errors(out, ref)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                K.gemm(A, B, trans_b=tb, bias=bias, out=out)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
        finally:
            so.valor_gemm_set_policy(7, o7); so.valor_gemm_set_policy(0, o0)
        bytes_ = 2.0 * (M * k + n * k + M * n)
        row[tag] = {"family": fam, "us": round(us, 1), "TBps": round(bytes_ / us / 1e6, 2), "rel_err": round(whole, 5), "worst_tile_err": round(worst, 5)}
        assert whole < 2.5e-3 and worst < 4e-3, (name, tag, whole, worst)
    res[name] = row
    print(name, row, flush=True)
json.dump(res, open(sys.argv[1], "w"), indent=1)
