"""The one-launch cross-attention forward / backward of all decoder passes of a layer (csrc/attention_xu.hip, valor_cross_attn_fwd_fused /
valor_cross_attn_bwd_fused) against
explicit softmax attention in fp64 AND against the per-pass kernels it replaces (valor_attn_bwd with dK|dV accumulation): the decoder's
geometry -- 3 caption groups x 32 rows with (start, len) key ranges + 42 mlm rows over 1834 keys shared per K/V batch, bert.py:448-457 --
plus ragged cases (a query tail, key counts that are not tile multiples, a group whose range starts inside a tile), with and without
dropout (same (seed, offset) windows as the forward: the regenerated masks must be the forward's)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / max(b.norm().item(), 5e-2 * b.numel() ** 0.5)).item()


def _ref(q, k, v, H, kv_range, bmod, scale):
    B, Sq, E = q.shape
    outs = []
    for b in range(B):
        kb = b % bmod
        s0, ln = (0, k.shape[1]) if kv_range is None else (int(kv_range[b, 0]), int(kv_range[b, 1]))
        qq = q[b].view(Sq, H, 64).transpose(0, 1)
        kk = k[kb, s0:s0 + ln].view(ln, H, 64).transpose(0, 1)
        vv = v[kb, s0:s0 + ln].view(ln, H, 64).transpose(0, 1)
        p = torch.softmax(qq @ kk.transpose(1, 2) * scale, -1)
        outs.append((p @ vv).transpose(0, 1).reshape(Sq, E))
    return torch.stack(outs)


# bmod, H, Skv, [(groups, T, ranges per group or None)]
CASES = [
    (4, 12, 1834, [(3, 32, [(0, 1834), (0, 1576), (1576, 258)]), (1, 42, None)]),      # the decoder: caption tva / tv / ta + mlm (3 x 2 + 3 sub-tiles)
    (3, 2, 330, [(3, 32, [(0, 330), (0, 200), (200, 130)]), (1, 42, None)]),            # ranges that start / end inside a 64-key tile, ragged last tile
    (2, 2, 64, [(1, 20, None)]),                                                       # one pass, one tile, a query tail
    (2, 3, 200, [(2, 48, [(0, 200), (37, 101)]), (2, 16, [(0, 64), (64, 136)])]),       # two grouped passes, 2 x 3 + 2 x 1 sub-tiles
    (1, 1, 1000, [(1, 96, None), (1, 64, None)]),                                      # ten sub-tiles
]


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("case", CASES)
def test_fused_backward_matches_fp64_and_the_per_pass_kernels(dev, case, p_drop):
    from valor_amd import kernels as K
    bmod, H, Skv, passes = case
    E = H * 64
    g = torch.Generator().manual_seed(Skv + 7 * bmod)
    scale = 1.0 / math.sqrt(64)
    kv = (torch.randn((bmod, Skv, 2 * E), generator=g) * 0.8).bfloat16().to(dev)
    k, v = kv[:, :, :E], kv[:, :, E:]
    segs = []
    for i, (G, T, ranges) in enumerate(passes):
        B = G * bmod
        q = (torch.randn((B, T, E), generator=g) * 0.8).bfloat16().to(dev)
        do = torch.randn((B, T, E), generator=g).bfloat16().to(dev)
        kvr = None
        if ranges is not None:
            kvr = torch.tensor([list(ranges[b // bmod]) for b in range(B)], dtype=torch.int32)
        seed, off = 11 + i, 1000 * (i + 1)
        o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr.to(dev) if kvr is not None else None, kv_bmod=bmod, scale=scale, p_drop=p_drop, seed=seed, offset=off)
        segs.append(dict(q=q, o=o, lse=lse, dout=do, kv_range=kvr.to(dev) if kvr is not None else None, kvr_cpu=kvr, seed=seed, offset=off))
    # per-pass kernels: first pass writes dK|dV, the second accumulates
    dkv_old = torch.empty_like(kv)
    dq_old = []
    for i, sg in enumerate(segs):
        dq, _, _ = K.attn_bwd(sg["q"], k, v, sg["o"], sg["lse"], sg["dout"], H, dk=dkv_old[:, :, :E], dv=dkv_old[:, :, E:], kv_range=sg["kv_range"],
                              kv_bmod=bmod, scale=scale, p_drop=p_drop, seed=sg["seed"], offset=sg["offset"], accumulate_kv=i > 0)
        dq_old.append(dq)
    # fused
    dkv_new = torch.full_like(kv, float("nan"))          # must be overwritten everywhere
    for sg in segs:
        sg["dq"] = torch.full_like(sg["q"], float("nan"))
    assert K.cross_attn_bwd_fused(segs, k, v, dkv_new[:, :, :E], dkv_new[:, :, E:], H, bmod, scale=scale, p_drop=p_drop)
    torch.cuda.synchronize()
    assert torch.isfinite(dkv_new.float()).all()
    for sg, dqo in zip(segs, dq_old):
        assert torch.isfinite(sg["dq"].float()).all()
        assert _rel(sg["dq"], dqo) < 1.2e-2, ("dq vs per-pass", _rel(sg["dq"], dqo))
    assert _rel(dkv_new, dkv_old) < 1.2e-2, ("dkv vs per-pass", _rel(dkv_new, dkv_old))
    if p_drop == 0.0:
        kd, vd = k.double().detach().requires_grad_(True), v.double().detach().requires_grad_(True)
        loss = 0
        qds = []
        for sg in segs:
            qd = sg["q"].double().detach().requires_grad_(True)
            qds.append(qd)
            loss = loss + (_ref(qd, kd, vd, H, sg["kvr_cpu"], bmod, scale) * sg["dout"].double()).sum()
        loss.backward()
        for sg, qd in zip(segs, qds):
            assert _rel(sg["dq"], qd.grad) < 2e-2, ("dq", _rel(sg["dq"], qd.grad))
        assert _rel(dkv_new[:, :, :E], kd.grad) < 2e-2 and _rel(dkv_new[:, :, E:], vd.grad) < 2e-2
        # one fp32 sum rounded once is no worse than two bf16 passes
        assert _rel(dkv_new[:, :, :E], kd.grad) <= _rel(dkv_old[:, :, :E], kd.grad) * 1.05 + 1e-4


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("case", CASES)
def test_fused_forward_matches_fp64_and_the_per_pass_kernels(dev, case, p_drop):
    from valor_amd import kernels as K
    bmod, H, Skv, passes = case
    E = H * 64
    g = torch.Generator().manual_seed(Skv + 7 * bmod + 1)
    scale = 1.0 / math.sqrt(64)
    kv = (torch.randn((bmod, Skv, 2 * E), generator=g) * 0.8).bfloat16().to(dev)
    k, v = kv[:, :, :E], kv[:, :, E:]
    segs, old = [], []
    for i, (G, T, ranges) in enumerate(passes):
        B = G * bmod
        q = (torch.randn((B, T, E), generator=g) * 1.5).bfloat16().to(dev)
        kvr = None
        if ranges is not None:
            kvr = torch.tensor([list(ranges[b // bmod]) for b in range(B)], dtype=torch.int32)
        seed, off = 21 + i, 777 * (i + 1)
        kd = kvr.to(dev) if kvr is not None else None
        old.append(K.attn_fwd(q, k, v, H, kv_range=kd, kv_bmod=bmod, scale=scale, p_drop=p_drop, seed=seed, offset=off))
        segs.append(dict(q=q, o=torch.full_like(q, float("nan")), lse=torch.full((B, H, T), float("nan"), device=dev), kv_range=kd, kvr_cpu=kvr,
                         seed=seed, offset=off))
    assert K.cross_attn_fwd_fused(segs, k, v, H, bmod, scale=scale, p_drop=p_drop)
    torch.cuda.synchronize()
    for sg, (o_old, lse_old) in zip(segs, old):
        assert torch.isfinite(sg["o"].float()).all() and torch.isfinite(sg["lse"]).all()
        # same keep pattern (p_drop > 0: the dropped entries are the per-pass kernel's) and the same softmax up to the order of the sums
        assert _rel(sg["o"], o_old) < 8e-3, ("o vs per-pass", _rel(sg["o"], o_old))
        assert (sg["lse"] - lse_old).abs().max().item() < 2e-3
        if p_drop == 0.0:
            ref = _ref(sg["q"].double(), k.double(), v.double(), H, sg["kvr_cpu"], bmod, scale)
            assert _rel(sg["o"], ref) < 8e-3, ("o", _rel(sg["o"], ref))
            assert _rel(sg["o"], ref) <= _rel(o_old, ref) * 1.1 + 1e-4


def test_fused_forward_and_backward_through_the_autograd_function(dev):
    """SegCrossAttnFn with the fused kernels on and off (VALOR_ATTN_XFUSED): same outputs and gradients on the decoder's geometry"""
    import os
    from valor_amd import ops
    bmod, H, Skv = 4, 12, 1834
    E = H * 64
    g = torch.Generator().manual_seed(5)
    kv0 = (torch.randn((bmod, Skv, 2 * E), generator=g) * 0.8).bfloat16().to(dev)
    q0 = (torch.randn((3 * bmod * 32 + bmod * 42, E), generator=g)).bfloat16().to(dev)
    do = torch.randn(q0.shape, generator=g).bfloat16().to(dev)
    kvr = torch.tensor([[(0, Skv), (0, 1576), (1576, 258)][r // bmod] for r in range(3 * bmod)], dtype=torch.int32).to(dev)
    segs = [(0, 3 * bmod, 32, kvr, bmod), (3 * bmod * 32, bmod, 42, None, bmod)]
    res = {}
    for mode in ("1", "0"):
        os.environ["VALOR_ATTN_XFUSED"] = mode
        try:
            q, kv = q0.clone().requires_grad_(True), kv0.clone().requires_grad_(True)
            o = ops.seg_cross_attention(q, kv, H, segs, 0.0)
            o.backward(do)
            res[mode] = (o.detach(), q.grad, kv.grad)
        finally:
            os.environ.pop("VALOR_ATTN_XFUSED", None)
    for a, b in zip(res["1"], res["0"]):
        assert _rel(a, b) < 1.2e-2


def test_outside_its_domain_the_fused_entry_declines(dev):
    from valor_amd import kernels as K
    E = 64
    kv = torch.randn((1, 128, 2 * E), device=dev).bfloat16()
    q = torch.randn((1, 200, E), device=dev).bfloat16()             # 13 sub-tiles > 10
    sg = dict(q=q, o=q, lse=torch.zeros((1, 1, 200), device=dev), dout=q, dq=torch.empty_like(q), kv_range=None)
    assert K.cross_attn_bwd_fused([sg], kv[:, :, :E], kv[:, :, E:], kv[:, :, :E].clone(), kv[:, :, E:].clone(), 1, 1) is False
    assert K.cross_attn_fwd_fused([sg], kv[:, :, :E], kv[:, :, E:], 1, 1) is False
