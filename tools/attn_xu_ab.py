"""cross-attention backward at the decoder's bench geometry (64 K/V batches x 12 heads, 1834 keys; caption 3 groups x 32 rows + mlm 42
rows, dropout 0.1): the two per-pass launches of csrc/attention_x.hip (second one accumulating dK|dV) against the one fused launch of
csrc/attention_xu.hip. usage: python tools/attn_xu_ab.py [out.json]"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K  # noqa: E402

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


def old():
    for i, sg in enumerate(segs):
        K.attn_bwd(sg["q"], k, v, sg["o"], sg["lse"], sg["dout"], H, dq=sg["dq"], dk=dkv[:, :, :E], dv=dkv[:, :, E:], kv_range=sg["kv_range"], kv_bmod=b,
                   scale=scale, p_drop=p, seed=sg["seed"], offset=sg["offset"], accumulate_kv=i > 0)


def new():
    assert K.cross_attn_bwd_fused(segs, k, v, dkv[:, :, :E], dkv[:, :, E:], H, b, scale=scale, p_drop=p)


def timeit(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fsegs = [dict(q=sg["q"], o=torch.empty_like(sg["o"]), lse=torch.empty_like(sg["lse"]), kv_range=sg["kv_range"], seed=sg["seed"], offset=sg["offset"])
         for sg in segs]


def old_fwd():
    for sg in fsegs:
        K.attn_fwd(sg["q"], k, v, H, kv_range=sg["kv_range"], kv_bmod=b, scale=scale, p_drop=p, seed=sg["seed"], offset=sg["offset"], o=sg["o"],
                   lse=sg["lse"])


def new_fwd():
    assert K.cross_attn_fwd_fused(fsegs, k, v, H, b, scale=scale, p_drop=p)


res = {"per_pass_us": [], "fused_us": [], "fwd_per_pass_us": [], "fwd_fused_us": []}
for _ in range(3):
    res["per_pass_us"].append(round(timeit(old), 1)); res["fused_us"].append(round(timeit(new), 1))
    res["fwd_per_pass_us"].append(round(timeit(old_fwd), 1)); res["fwd_fused_us"].append(round(timeit(new_fwd), 1))
res["fwd_necessary_GB"] = round((kv.numel() * 2 + sum(2 * sg["q"].numel() * 2 for sg in segs)) / 1e9, 3)
nec = (2 * kv.numel() * 2 + sum(4 * sg["q"].numel() * 2 for sg in segs)) / 1e9
res["necessary_GB"] = round(nec, 3)
res["fused_TBps_of_necessary"] = round(nec / min(res["fused_us"]) * 1e3, 2)
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
