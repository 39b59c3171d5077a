"""Run ONE valor_gemm shape a few times (for rocprofv3 --pmc passes). usage: gemm_one.py M N K ta tb [reps]"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K  # noqa: E402

M, N, Kd, ta, tb = [int(x) for x in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
A = torch.randn((Kd, M) if ta else (M, Kd), device=dev).bfloat16()
B = torch.randn((Kd, N) if tb else (N, Kd), device=dev).bfloat16()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(reps):
    K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
torch.cuda.synchronize()
