"""Self-attention backward at the short text shapes of the step (decoder caption pass 192 x 12 heads x 32 tokens with a causal mask and dropout, mlm
pass 64 x 12 x 42 with a padding mask and dropout, CLIP text tower 64 x 12 x 32 causal): the one-wave-per-head LDS-resident kernel
(attn_res_bwd1_kernel, default) against the streaming dQ + dK/dV pair (VALOR_ATTN_SHORT=0; the switch is read once per process, so two runs).
Prints HIP-event times and a checksum of the gradients; with a second argument compares against the gradients a previous run saved.
usage: python tools/attn_short_ab.py out.json [ref.pt]"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
H, E = 12, 768
scale = 1.0 / math.sqrt(64)
res, grads = {"VALOR_ATTN_SHORT": os.environ.get("VALOR_ATTN_SHORT", "1")}, {}
for name, B, S, p, kind in [("caption_192x32_causal_drop", 192, 32, 0.1, "causal"), ("mlm_64x42_padding_drop", 64, 42, 0.1, "pad"), ("clip_text_64x32_causal", 64, 32, 0.0, "causal")]:
    g = torch.Generator().manual_seed(S + B)
    qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
    if kind == "causal":
        mask = torch.triu(torch.full((S, S), -10000.0), diagonal=1)[None].contiguous().to(dev)
    else:
        lens = torch.randint(20, S + 1, (B,), generator=g)
        mask = torch.zeros((B, S, S))
        for b in range(B):
            mask[b, :, lens[b]:] = -10000.0
        mask = mask.to(dev)
    o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=scale, p_drop=p, seed=5, offset=9)
    dqkv = torch.empty_like(qkv)
    run = lambda: K.attn_bwd(q, k, v, o, lse, dout, H, dq=dqkv[:, :, :E], dk=dqkv[:, :, E:2 * E], dv=dqkv[:, :, 2 * E:], mask=mask, scale=scale, p_drop=p, seed=5, offset=9)
    run()
    grads[name] = dqkv.clone()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    fwd = lambda: K.attn_fwd(q, k, v, H, mask=mask, scale=scale, p_drop=p, seed=5, offset=9)
    tf = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fwd()
        e1.record(); torch.cuda.synchronize()
        tf.append(e0.elapsed_time(e1) / 20 * 1e3)
    grads[name + "_o"] = o.clone()
    res[name] = {"us": round(sorted(ts)[1], 1), "fwd_us": round(sorted(tf)[1], 1), "finite": bool(torch.isfinite(dqkv.float()).all())}
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    ref = torch.load(sys.argv[2])
    for n in grads:
        a, b = grads[n].double(), ref[n].double().to(dev)
        if n.endswith("_o"):
            res[n[:-2]]["fwd_rel_diff_vs_other_family"] = float((a - b).norm() / b.norm())
        else:
            res[n]["rel_diff_vs_other_family"] = float((a - b).norm() / b.norm())
elif len(sys.argv) > 2:
    torch.save({n: t.cpu() for n, t in grads.items()}, sys.argv[2])
print(json.dumps(res))
json.dump(res, open(sys.argv[1], "w"), indent=1)
