"""A/B of the cross-attention kernels (valor_attn_set_variant bit 1): bf16 correctness vs an fp64 reference with
grouped kv ranges, dropout mask agreement across variants / fwd-bwd, and timings on the VALOR-base decoder shapes.
usage: python tools/attn_x_ab.py [variants e.g. 0,3]"""
import math
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from valor_amd import kernels as K, lib  # noqa: E402
from test_attention_gpu import _ref_attn, _rel  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
VARS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,3".split(","))]
scale = 1.0 / math.sqrt(64)


def make(B, H, Sq, Skv, bmod, ranges, seed):
    g = torch.Generator().manual_seed(seed)
    E = H * 64
    Bkv = bmod if bmod > 0 else B
    q = (torch.randn((B, Sq, 2 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)[:, :, :E]
    kvb = (torch.randn((Bkv, Skv, 2 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    k, v = kvb[:, :, :E], kvb[:, :, E:]
    kvr = None
    if ranges is not None:
        kvr = torch.tensor([list(ranges[b // bmod]) for b in range(B)], dtype=torch.int32)
    dout = torch.randn((B, Sq, E), generator=g).to(torch.bfloat16).to(dev)
    return q, k, v, kvr, dout


def check():
    cases = [
        (2, 12, 32, 458, 0, None),
        (6, 12, 32, 330, 2, [(0, 330), (0, 200), (200, 130)]),
        (6, 3, 32, 1834, 2, [(0, 1834), (0, 1576), (1576, 258)]),
        (2, 3, 42, 1834, 2, [(0, 1834)]),
        (4, 2, 48, 700, 2, [(0, 700), (130, 333)]),
        (3, 2, 17, 129, 3, [(5, 100)]),
        (8, 2, 16, 300, 2, [(0, 300), (0, 10), (290, 10), (100, 100)]),
    ]
    for (B, H, Sq, Skv, bmod, ranges) in cases:
        q, k, v, kvr, dout = make(B, H, Sq, Skv, bmod, ranges, 5 + Sq + Skv)
        qd, kd, vd = (t.double().detach().requires_grad_(True) for t in (q, k, v))
        oref = _ref_attn(qd, kd, vd, H, None, kvr, bmod, scale)
        (oref * dout.double()).sum().backward()
        kvr_d = kvr.to(dev) if kvr is not None else None
        for var in VARS:
            so.valor_attn_set_variant(var)
            o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr_d, kv_bmod=bmod, scale=scale)
            dq, dk, dv = K.attn_bwd(q, k, v, o, lse, dout, H, kv_range=kvr_d, kv_bmod=bmod, scale=scale)
            errs = (_rel(o, oref), _rel(dq, qd.grad), _rel(dk, kd.grad), _rel(dv, vd.grad))
            ok = errs[0] < 1e-2 and max(errs[1:]) < 2e-2
            print(f"variant {var} B={B} H={H} Sq={Sq} Skv={Skv} bmod={bmod}: o {errs[0]:.2e} dq {errs[1]:.2e} dk {errs[2]:.2e} dv {errs[3]:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    # dropout agreement: q = 0 -> uniform P over the range; v one-hot in the first 64 keys of the range
    B, H, Sq, Skv, bmod, pd = 4, 2, 32, 256, 2, 0.25
    E = H * 64
    q = torch.zeros((B, Sq, E), device=dev, dtype=torch.bfloat16)
    k = torch.randn((bmod, Skv, E), device=dev).to(torch.bfloat16)
    v = torch.eye(Skv, device=dev)[:, :64].repeat(1, H)[None].expand(bmod, Skv, E).contiguous().to(torch.bfloat16)
    kvr = torch.tensor([[0, 256], [0, 256], [0, 128], [0, 128]], dtype=torch.int32).to(dev)
    outs = {}
    for var in VARS:
        so.valor_attn_set_variant(var)
        o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr, kv_bmod=bmod, p_drop=pd, seed=7, offset=11)
        n = torch.tensor([256, 256, 128, 128], device=dev).view(B, 1, 1).float()
        keep = (o.float() * n * (1 - pd) > 0.5)
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, torch.ones_like(o), H, kv_range=kvr, kv_bmod=bmod, p_drop=pd, seed=7, offset=11)
        # dV[kvb, key<64, h, 0] = sum over query batches mapping to kvb, rows q: keep/(1-p)/n
        want = (keep.float() / (1 - pd) / n).view(B, Sq, H, 64).sum(1)                     # [B,H,64]
        want = want.view(B // bmod, bmod, H, 64).sum(0)                                    # [bmod,H,64]
        got = dv.float().view(bmod, Skv, H, 64)[:, :64, :, 0].permute(0, 2, 1)
        outs[var] = keep
        print(f"variant {var} dropout: keep frac {keep.float().mean().item():.4f} (want {1 - pd}), fwd/bwd agreement max err {(got - want).abs().max().item():.3e}", flush=True)
    if len(VARS) > 1:
        print("dropout masks identical across variants:", bool(torch.equal(outs[VARS[0]], outs[VARS[1]])), flush=True)


def bench(rounds=3, n=5):
    for name, B, Sq, bmod, ranges, pd in [("caption", 192, 32, 64, [(0, 1834), (0, 1576), (1576, 258)], 0.1),
                                          ("mlm", 64, 42, 64, [(0, 1834)], 0.1),
                                          ("caption_nodrop", 192, 32, 64, [(0, 1834), (0, 1576), (1576, 258)], 0.0)]:
        H, Skv = 12, 1834
        q, k, v, kvr, dout = make(B, H, Sq, Skv, bmod, ranges, 3)
        kvr = kvr.to(dev)
        fl = sum(4.0 * bmod * H * Sq * ln * 64 for (_, ln) in ranges)
        res = {}
        for var in VARS:
            bf = bb = 1e9
            for r in range(rounds):
                so.valor_attn_set_variant(var)
                o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr, kv_bmod=bmod, scale=scale, p_drop=pd, seed=1, offset=0)
                K.attn_bwd(q, k, v, o, lse, dout, H, kv_range=kvr, kv_bmod=bmod, scale=scale, p_drop=pd, seed=1, offset=0)
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                for _ in range(n):
                    o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr, kv_bmod=bmod, scale=scale, p_drop=pd, seed=1, offset=0)
                e[1].record()
                for _ in range(n):
                    K.attn_bwd(q, k, v, o, lse, dout, H, kv_range=kvr, kv_bmod=bmod, scale=scale, p_drop=pd, seed=1, offset=0)
                e[2].record()
                torch.cuda.synchronize()
                bf = min(bf, e[0].elapsed_time(e[1]) / n); bb = min(bb, e[1].elapsed_time(e[2]) / n)
            res[var] = (bf, bb)
        print(f"{name:14s} B={B} Sq={Sq} p={pd}: " + "  ".join(f"v{var}: fwd {res[var][0]*1e3:7.1f} us ({fl/res[var][0]/1e9:6.1f} TF)  bwd {res[var][1]*1e3:7.1f} us ({2.5*fl/res[var][1]/1e9:6.1f} TF)" for var in VARS), flush=True)


if __name__ == "__main__":
    check()
    bench()
