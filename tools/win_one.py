"""One VideoSwin window-attention forward + backward at a production shape (for rocprofv3 --pmc passes).
usage: win_one.py [stage] [reps]   stage 0..3 of VideoSwin-B at per-GPU batch 64, 8 frames (C = 128 << stage)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from valor_amd import kernels as K, synth  # noqa: E402
from valor_amd.model.valor import VALOR  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
spec = synth.tiny_swin_spec()
m = VALOR({"dropout": 0.0}, spec=spec, dtype=torch.bfloat16, device=dev)
B, heads, side = 64, 4 << stage, 56 >> stage
C = heads * 32
D = 8
for shifted in (False, True):
    geo = m._swin_geometry(D, side, side, shifted)
    qkv = torch.randn((B * D * side * side, 3 * C), device=dev, dtype=torch.bfloat16)
    table = (0.02 * torch.randn((spec.swin_table, heads), device=dev)).to(torch.bfloat16)
    do = torch.randn((qkv.shape[0], C), device=dev, dtype=torch.bfloat16)
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o, lse = K.win_attn_fwd(qkv, geo, table, heads, B)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        K.win_attn_bwd(qkv, o, lse, do, geo, table, heads, B)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    pairs = B * geo["nW"] * heads
    fl = 4.0 * geo["N"] ** 2 * 32 * pairs
    print(f"stage {stage} shifted={shifted} windows x heads = {pairs} N={geo['N']}: fwd {1e3*(t1-t0):.3f} ms ({fl/(t1-t0)/1e12:.0f} TF/s), "
          f"bwd {1e3*(t2-t1):.3f} ms ({2.5*fl/(t2-t1)/1e12:.0f} TF/s)")
