"""A/B of attention kernel variants (valor_attn_set_variant): bf16 correctness vs an fp64 reference (incl. masks and
dropout fwd/bwd consistency) and HIP-event timings on the VALOR-base self-attention shapes.
usage: python tools/attn_ab.py [variants e.g. 0,1]"""
import math
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from valor_amd import kernels as K, lib  # noqa: E402
from test_attention_gpu import _ref_attn, _rel  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
VARS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1".split(","))]
scale = 1.0 / math.sqrt(64)


def case(B, H, S, masked, seed):
    g = torch.Generator().manual_seed(seed)
    E = H * 64
    qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    mask = None
    if masked:
        lens = torch.randint(3, S + 1, (B,), generator=g)
        m = (torch.arange(S)[None, :] < lens[:, None]).float()[:, None, :].expand(B, S, S).clone()
        mask = ((1.0 - torch.tril(m)) * -10000.0).to(dev).contiguous()
    dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
    return q, k, v, mask, dout


def check():
    for (B, H, S, masked) in [(2, 12, 197, False), (3, 12, 129, False), (4, 12, 32, True), (2, 12, 42, True), (2, 8, 32, True),
                              (1, 2, 1, False), (2, 3, 256, False), (2, 2, 17, True), (1, 1, 64, False), (2, 2, 100, False)]:
        q, k, v, mask, dout = case(B, H, S, masked, 100 + S)
        qd, kd, vd = (t.double().detach().requires_grad_(True) for t in (q, k, v))
        oref = _ref_attn(qd, kd, vd, H, mask, None, 0, scale)
        (oref * dout.double()).sum().backward()
        for var in VARS:
            so.valor_attn_set_variant(var)
            o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=scale)
            dq, dk, dv = K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, scale=scale)
            errs = (_rel(o, oref), _rel(dq, qd.grad), _rel(dk, kd.grad), _rel(dv, vd.grad))
            ok = errs[0] < 1e-2 and max(errs[1:]) < 2e-2
            print(f"variant {var} B={B} H={H} S={S} masked={masked}: o {errs[0]:.2e} dq {errs[1]:.2e} dk {errs[2]:.2e} dv {errs[3]:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    # dropout: same (seed, offset) window -> the variants must produce the same mask (bitwise-equal keep pattern),
    # and fwd/bwd of one variant must agree: with q = 0 (uniform P) and v = one-hot rows, o recovers the keep mask and
    # dV[key, 0] = sum_q keep(q, key) / (1-p) / S.
    B, H, S, pd = 2, 2, 128, 0.25
    E = H * 64
    q = torch.zeros((B, S, E), device=dev, dtype=torch.bfloat16)
    k = torch.randn((B, S, E), device=dev).to(torch.bfloat16)
    v = torch.eye(S, device=dev)[:, :64].repeat(1, H)[None].expand(B, S, E).contiguous().to(torch.bfloat16)
    outs = {}
    for var in VARS:
        so.valor_attn_set_variant(var)
        o, lse = K.attn_fwd(q, k, v, H, p_drop=pd, seed=7, offset=11)
        keep = (o.float() * S * (1 - pd) > 0.5)
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, torch.ones_like(o), H, p_drop=pd, seed=7, offset=11)
        want = keep.float().view(B, S, H, 64).sum(1) / (1 - pd) / S
        got = dv.float().view(B, S, H, 64)[:, :64, :, 0].permute(0, 2, 1)
        outs[var] = keep
        print(f"variant {var} dropout: keep frac {keep.float().mean().item():.4f} (want {1 - pd}), fwd/bwd mask agreement max err {(got - want).abs().max().item():.3e}", flush=True)
    if len(VARS) > 1:
        print("dropout masks identical across variants:", all(bool(torch.equal(outs[VARS[0]], outs[v])) for v in VARS[1:]), flush=True)


def bench(rounds=3, n=5):
    for name, B, H, S, masked, pd in [("vit", 512, 12, 197, False, 0.0), ("ast", 128, 12, 129, False, 0.1), ("dec_self", 192, 12, 32, True, 0.1),
                                      ("mlm_self", 64, 12, 42, True, 0.1)]:
        q, k, v, mask, dout = case(B, H, S, masked, 7)
        fl = 4.0 * B * H * S * S * 64
        res = {}
        for var in VARS:
            bf = bb = 1e9
            for r in range(rounds):
                so.valor_attn_set_variant(var)
                o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=scale, p_drop=pd, seed=1, offset=0)
                K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, scale=scale, p_drop=pd, seed=1, offset=0)
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                for _ in range(n):
                    o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=scale, p_drop=pd, seed=1, offset=0)
                e[1].record()
                for _ in range(n):
                    K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, scale=scale, p_drop=pd, seed=1, offset=0)
                e[2].record()
                torch.cuda.synchronize()
                bf = min(bf, e[0].elapsed_time(e[1]) / n); bb = min(bb, e[1].elapsed_time(e[2]) / n)
            res[var] = (bf, bb)
        print(f"{name:9s} B={B} S={S} p={pd}: " + "  ".join(f"v{var}: fwd {res[var][0]*1e3:7.1f} us ({fl/res[var][0]/1e9:6.1f} TF)  bwd {res[var][1]*1e3:7.1f} us ({2.5*fl/res[var][1]/1e9:6.1f} TF)" for var in VARS), flush=True)


if __name__ == "__main__":
    check()
    bench()
