"""ONE launch of every non-GEMM hot kernel of the VALOR-base step at its bench shape (per-GPU batch 64), for rocprofv3 --pmc passes
(tools/gpu_pmc_kernels.sh): LDS-resident self-attention forward / backward (ViT 512 x 12 heads x S = 197; AST with dropout),
key-stationary cross-attention forward / backward (caption pass: 3 groups x 32 rows against 1834 keys, dropout; the mlm pass; both passes in
one launch), fused LayerNorm forward /
backward, cross-entropy, fused AdamW, the fused contrastive forward.
usage: python tools/kernels_one.py [reps=1]            one launch each, for the counter passes
       python tools/kernels_one.py time out.json       HIP-event time of each (median of 3 rounds x 5 launches); VALOR_HIP_LIB selects the
                                                       library build, so two runs A/B a kernel change on one box"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402
from valor_amd.kernels import _ptr, _stream  # noqa: E402

TIME = len(sys.argv) > 2 and sys.argv[1] == "time"
reps = 1 if TIME else (int(sys.argv[1]) if len(sys.argv) > 1 else 1)
NAMES = ["self_fwd_vit", "self_bwd_vit", "self_fwd_ast_drop", "self_bwd_ast_drop", "cross_fwd_caption", "cross_bwd_caption", "cross_fwd_mlm", "cross_bwd_mlm_acc",
         "cross_bwd_fused", "cross_fwd_fused", "ln_fwd", "ln_bwd", "xent_fwd", "xent_bwd", "adamw", "fine_fused_fwd"]
dev = torch.device("cuda:0")
scale = 1.0 / math.sqrt(64)
g = torch.Generator().manual_seed(1)
H = 12
E = H * 64
todo = []

# ---- self-attention (ViT / AST)
for B, S, p in ((512, 197, 0.0), (128, 129, 0.1)):
    qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
    o, lse = K.attn_fwd(q, k, v, H, scale=scale, p_drop=p, seed=5, offset=9)
    dqkv = torch.empty_like(qkv)
    todo.append(lambda q=q, k=k, v=v, p=p: K.attn_fwd(q, k, v, H, scale=scale, p_drop=p, seed=5, offset=9))
    todo.append(lambda q=q, k=k, v=v, o=o, lse=lse, dout=dout, d=dqkv, p=p: K.attn_bwd(q, k, v, o, lse, dout, H, dq=d[:, :, :E], dk=d[:, :, E:2 * E], dv=d[:, :, 2 * E:],
                                                                                         scale=scale, p_drop=p, seed=5, offset=9))
# ---- cross-attention: caption pass (192 query batches = 3 groups x 64 clips, 32 rows) and mlm pass (64 x 42 rows), 1834 keys, dropout 0.1
kvb = (torch.randn((64, 1834, 2 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
kk, vv = kvb[:, :, :E], kvb[:, :, E:]
dkv = torch.empty_like(kvb)
xsegs = []
for Bq, Sq, ranges in ((192, 32, [(0, 1834), (0, 1576), (1576, 258)]), (64, 42, [(0, 1834)])):
    q = (torch.randn((Bq, Sq, E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    kvr = torch.tensor([list(ranges[b // 64]) for b in range(Bq)], dtype=torch.int32).to(dev)
    dout = torch.randn((Bq, Sq, E), generator=g).to(torch.bfloat16).to(dev)
    o, lse = K.attn_fwd(q, kk, vv, H, kv_range=kvr, kv_bmod=64, scale=scale, p_drop=0.1, seed=3, offset=1)
    todo.append(lambda q=q, kvr=kvr: K.attn_fwd(q, kk, vv, H, kv_range=kvr, kv_bmod=64, scale=scale, p_drop=0.1, seed=3, offset=1))
    todo.append(lambda q=q, kvr=kvr, o=o, lse=lse, dout=dout, acc=(Sq == 42): K.attn_bwd(q, kk, vv, o, lse, dout, H, dk=dkv[:, :, :E], dv=dkv[:, :, E:], kv_range=kvr, kv_bmod=64,
                                                                                          scale=scale, p_drop=0.1, seed=3, offset=1, accumulate_kv=acc))
    xsegs.append(dict(q=q, o=o, lse=lse, dout=dout, dq=torch.empty_like(q), kv_range=kvr, seed=3, offset=1))
# the same two passes in ONE launch (csrc/attention_xu.hip): what the step runs in backward; the forward twin is opt-in
todo.append(lambda: K.cross_attn_bwd_fused(xsegs, kk, vv, dkv[:, :, :E], dkv[:, :, E:], H, 64, scale=scale, p_drop=0.1))
fsegs = [dict(q=sg["q"], o=torch.empty_like(sg["o"]), lse=torch.empty_like(sg["lse"]), kv_range=sg["kv_range"], seed=3, offset=1) for sg in xsegs]
todo.append(lambda: K.cross_attn_fwd_fused(fsegs, kk, vv, H, 64, scale=scale, p_drop=0.1))
# ---- fused LayerNorm at the ViT shape
rows, cols = 100864, 768
x = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev); r = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev)
bias, gam, bet = (torch.randn(cols, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
dy, dz = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev), torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev)
z, y, mean, rstd = K.bdrln_fwd(x, bias, r, gam, bet, 1e-5)
todo.append(lambda: K.bdrln_fwd(x, bias, r, gam, bet, 1e-5))
todo.append(lambda: K.bdrln_bwd(dy, dz, z, mean, rstd, gam, want_dbias=True))
# ---- cross-entropy over the padded vocabulary
n, V, Vpad = 2100, 30522, 30528
logits = (3.0 * torch.randn((n, Vpad), generator=g)).to(torch.bfloat16).to(dev)
labels = torch.randint(0, V, (n,), generator=g).to(dev)
loss_rows = torch.empty(n, device=dev); lse_x = torch.empty(n, device=dev); gs = torch.ones((), device=dev)
todo.append(lambda: lib.call("valor_xent_fwd", _stream(), 0, _ptr(logits), _ptr(labels), _ptr(loss_rows), _ptr(lse_x), n, V, Vpad))
todo.append(lambda: lib.call("valor_xent_bwd", _stream(), 0, _ptr(logits), _ptr(labels), _ptr(lse_x), _ptr(gs), 1.0 / n, n, V, Vpad))
# ---- AdamW over the 374.7 M parameters
N = 374_784_000 // 1024 * 1024
master = torch.randn(N, device=dev); m1 = torch.zeros(N, device=dev); v1 = torch.zeros(N, device=dev)
grad = torch.zeros(N, device=dev, dtype=torch.bfloat16); param = torch.zeros(N, device=dev, dtype=torch.bfloat16)
table = torch.zeros(N // 1024, dtype=torch.int8, device=dev)
lr = (ctypes.c_float * 10)(*([1e-4] * 10)); wd = (ctypes.c_float * 10)(*([0.01] * 10))
todo.append(lambda: lib.call("valor_adamw", _stream(), 0, _ptr(master), _ptr(m1), _ptr(v1), _ptr(grad), _ptr(param), _ptr(table), N, lr, wd, 10,
                             0.9, 0.98, 1e-6, 3, 1, _ptr(gs), 1))
# ---- fused contrastive forward at the global batch of 8 ranks
Bc, T, Nv, D = 512, 32, 10, 512
fa = torch.nn.functional.normalize(torch.randn((Bc, T, D), generator=g), dim=-1).to(torch.bfloat16).to(dev)
fb = torch.nn.functional.normalize(torch.randn((Bc, Nv, D), generator=g), dim=-1).to(torch.bfloat16).to(dev)
mA, mB = torch.ones((Bc, T), device=dev), torch.ones((Bc, Nv), device=dev)
wA, wB = torch.full((Bc, T), 1.0 / T, device=dev), torch.full((Bc, Nv), 1.0 / Nv, device=dev)
sc = torch.empty((Bc, Bc), device=dev); a2b = torch.empty((Bc, Bc, T), device=dev); b2a = torch.empty((Bc, Bc, Nv), device=dev)
ia = torch.empty((Bc, Bc, T), dtype=torch.uint8, device=dev); ib = torch.empty((Bc, Bc, Nv), dtype=torch.uint8, device=dev)
todo.append(lambda: lib.call("valor_fine_fused_fwd", _stream(), _ptr(fa), _ptr(fb), _ptr(mA), _ptr(mB), _ptr(wA), _ptr(wB), _ptr(sc), _ptr(a2b), _ptr(b2a),
                             _ptr(ia), _ptr(ib), Bc, Bc, T, Nv, D))
torch.cuda.synchronize()
if TIME:
    import json
    assert len(NAMES) == len(todo)
    t = {name: [] for name in NAMES}
    for rnd in range(3):
        for name, f in zip(NAMES, todo):          # (the launch lambdas read the module-level n, V, ...: no loop variable may shadow them)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            t[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    res = {name: round(sorted(v)[1], 1) for name, v in t.items()}
    res["library"] = os.path.basename(lib.LIB_PATH)
    print(res, flush=True)
    json.dump(res, open(sys.argv[2], "w"), indent=1)
    sys.exit(0)
for _ in range(reps):
    for f in todo:
        f()
        torch.cuda.synchronize()
