"""GPU parity of the flash attention kernels (fwd + bwd) against explicit softmax attention in fp64."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    # relative L2 error; a reference that is (numerically) zero falls back to an absolute scale of 1e-3/elt
    a, b = a.double(), b.double()
    return ((a - b).norm() / max(b.norm().item(), 5e-2 * b.numel() ** 0.5)).item()


def _ref_attn(q, k, v, H, mask, kv_range, kv_bmod, scale):
    """q [B,Sq,E], k/v [Bkv,Skv,E] (fp64, requires_grad). Returns o [B,Sq,E]."""
    B, Sq, E = q.shape
    outs = []
    for b in range(B):
        kb = b % kv_bmod if kv_bmod > 0 else b
        s0, ln = (0, k.shape[1]) if kv_range is None else (int(kv_range[b, 0]), int(kv_range[b, 1]))
        qq = q[b].view(Sq, H, 64).transpose(0, 1)
        kk = k[kb, s0:s0 + ln].view(ln, H, 64).transpose(0, 1)
        vv = v[kb, s0:s0 + ln].view(ln, H, 64).transpose(0, 1)
        s = qq @ kk.transpose(1, 2) * scale
        if mask is not None:
            s = s + mask[b if mask.shape[0] > 1 else 0, :, :ln].double()
        p = torch.softmax(s, -1)
        outs.append((p @ vv).transpose(0, 1).reshape(Sq, E))
    return torch.stack(outs)


CASES = [
    # B, H, Sq, Skv, masked, ranged
    (2, 12, 197, 197, False, False),    # CLIP ViT-B/16 frame
    (3, 12, 129, 129, False, False),    # AST slice
    (4, 12, 32, 32, True, False),       # BERT text (pad + causal additive mask)
    (2, 12, 42, 42, True, False),       # text + task prompt
    (2, 8, 32, 32, True, False),        # CLIP text (8 heads)
    (3, 12, 100, 100, True, False),     # additive mask on the LDS-resident backward (S > 64: its <., MASK> instantiations; no model path uses them)
    (2, 12, 32, 458, False, False),     # cross-attention, ragged tile tail
    (6, 12, 32, 330, False, True),      # modality-grouped: 3 query groups share K/V of batch 2
    (1, 2, 1, 1, False, False),
    (1, 1, 70, 64, False, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_attention_fwd_bwd(dev, dtype, case):
    from valor_amd import kernels as K
    B, H, Sq, Skv, masked, ranged = case
    E = H * 64
    g = torch.Generator().manual_seed(B * 1000 + Sq + Skv)
    kv_bmod = 0
    Bkv = B
    kv_range = None
    if ranged:
        kv_bmod = 2
        Bkv = 2
        # group 0: all rows, group 1: first 200 ("video"), group 2: the rest ("audio")
        kv_range = torch.tensor([[0, Skv], [0, Skv], [0, 200], [0, 200], [200, Skv - 200], [200, Skv - 200]], dtype=torch.int32)
    # fused-QKV style storage: q/k/v are column slices of wider buffers (strided rows)
    qkv = (torch.randn((B, Sq, 3 * E), generator=g) * 0.8).to(dtype).to(dev)
    kvb = (torch.randn((Bkv, Skv, 2 * E), generator=g) * 0.8).to(dtype).to(dev)
    q = qkv[:, :, :E]
    if Sq == Skv and not ranged:
        k, v = qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    else:
        k, v = kvb[:, :, :E], kvb[:, :, E:]
    mask = None
    if masked:
        lens = torch.randint(3, Sq + 1, (B,), generator=g)
        m = (torch.arange(Sq)[None, :] < lens[:, None]).float()
        m = m[:, None, :].expand(B, Sq, Sq).clone()
        m = torch.tril(m)
        mask = ((1.0 - m) * -10000.0).to(dev).contiguous()
    dout = (torch.randn((B, Sq, E), generator=g)).to(dtype).to(dev)
    scale = 1.0 / math.sqrt(64)
    kvr_dev = kv_range.to(dev) if kv_range is not None else None

    o, lse = K.attn_fwd(q, k, v, H, mask=mask, kv_range=kvr_dev, kv_bmod=kv_bmod, scale=scale)
    dq, dk, dv = K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, kv_range=kvr_dev, kv_bmod=kv_bmod, scale=scale)

    qd = q.double().detach().requires_grad_(True)
    kd = k.double().detach().requires_grad_(True)
    vd = v.double().detach().requires_grad_(True)
    oref = _ref_attn(qd, kd, vd, H, mask, kv_range, kv_bmod, scale)
    (oref * dout.double()).sum().backward()
    tol_f, tol_b = (3e-6, 1e-5) if dtype == torch.float32 else (1e-2, 2e-2)
    assert _rel(o, oref) < tol_f, ("o", _rel(o, oref))
    assert _rel(dq, qd.grad) < tol_b, ("dq", _rel(dq, qd.grad))
    assert _rel(dk, kd.grad) < tol_b, ("dk", _rel(dk, kd.grad))
    assert _rel(dv, vd.grad) < tol_b, ("dv", _rel(dv, vd.grad))


def test_attention_dropout(dev):
    """dropout keeps ~1-p of the probabilities, is reproducible, and fwd/bwd use the same mask
    (checked through the identity dV = P_drop^T dO with V-independent P when q = 0)."""
    from valor_amd import kernels as K
    B, H, Sq, Skv, p = 2, 2, 64, 128, 0.25
    E = H * 64
    q = torch.zeros((B, Sq, E), device=dev)                     # uniform softmax: P = 1/Skv
    k = torch.randn((B, Skv, E), device=dev)
    v = torch.eye(Skv, device=dev)[:, :64].repeat(1, H)[None].expand(B, Skv, E).contiguous()  # v[key] = onehot(key) for key<64
    o, lse = K.attn_fwd(q, k, v, H, p_drop=p, seed=7, offset=11)
    # o[b,q,h*64+d] = keep(q,key=d)/(1-p)/Skv  for d < 64
    keepmask = o * Skv * (1 - p)
    frac = (keepmask > 0.5).float().mean().item()
    assert abs(frac - (1 - p)) < 0.02
    o2, _ = K.attn_fwd(q, k, v, H, p_drop=p, seed=7, offset=11)
    assert torch.equal(o, o2)
    dout = torch.ones_like(o)
    dq, dk, dv = K.attn_bwd(q, k, v, o, lse, dout, H, p_drop=p, seed=7, offset=11)
    # dV[key, :] = sum_q Pdrop[q,key] * 1 ; for key < 64 compare with the mask recovered from o
    want = keepmask.view(B, Sq, H, 64).sum(1) / (1 - p) / Skv          # [B,H,key<64]
    got = dv.view(B, Skv, H, 64)[:, :64, :, 0].permute(0, 2, 1)        # dv[b,key,h,0]
    assert torch.allclose(got, want, atol=1e-5), (got - want).abs().max()


@pytest.mark.parametrize("B,S,p_drop,causal", [(24, 32, 0.1, True), (8, 42, 0.1, False), (8, 32, 0.0, True), (5, 64, 0.1, False), (3, 7, 0.1, True)])
def test_short_sequences_one_wave_per_head_backward(dev, B, S, p_drop, causal):
    """Sequences of up to 64 rows (the decoder passes' and the CLIP text tower's 32 / 42-token self-attention, bert.py:272-288, clip.py:407-414)
    take attn_res_bwd1_kernel -- one wave per (batch, head), dQ and dK / dV in one launch -- instead of the streaming dQ + dK/dV pair: the same
    gradients to bf16 rounding, incl. the regenerated dropout mask (a different mask would differ in O(1)) and additive masks."""
    from valor_amd import kernels as K, lib
    so = lib.load()
    H = 12
    E = H * 64
    g = torch.Generator().manual_seed(S * 3 + B)
    qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
    if causal:
        mask = torch.triu(torch.full((S, S), -10000.0), diagonal=1)[None].contiguous().to(dev)
    else:
        lens = torch.randint(max(1, S // 2), S + 1, (B,), generator=g)
        mask = torch.zeros((B, S, S))
        for b in range(B):
            mask[b, :, lens[b]:] = -10000.0
        mask = mask.to(dev)
    scale = 1.0 / math.sqrt(64)
    o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=scale, p_drop=p_drop, seed=5, offset=9)
    got = K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, scale=scale, p_drop=p_drop, seed=5, offset=9)
    old = so.valor_attn_set_variant(0)              # streaming kernels only
    try:
        want = K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, scale=scale, p_drop=p_drop, seed=5, offset=9)
    finally:
        so.valor_attn_set_variant(old)
    for a, b_, n in zip(got, want, ("dq", "dk", "dv")):
        assert torch.isfinite(a.float()).all(), n
        assert _rel(a, b_) < 2e-3, (n, _rel(a, b_))


@pytest.mark.parametrize("S,B,H,p_drop", [(197, 44, 12, 0.0), (197, 44, 12, 0.1), (129, 48, 12, 0.1), (256, 43, 12, 0.0), (65, 90, 6, 0.1), (224, 64, 8, 0.0)])
def test_resident_backward_pipelined_equals_per_head_kernel(dev, S, B, H, p_drop):
    """The persistent, phase-pipelined LDS-resident backward (attention_res.hip: one workgroup per CU walks (batch, head) items, the
    K / V and Q / dO image pairs double-buffered across its two phases) does the arithmetic of the one-workgroup-per-head kernel (only the
    16 products of a row's delta = sum(dO * O) are added in another order): the same dQ / dK / dV to an ulp of bf16, at item counts that give every workgroup several (and unequal numbers of) items, a tail
    round, sequence lengths with partial 32-row blocks, with and without dropout; and against fp64 on one of the shapes.
    Reference semantics: model/clip.py:186-192 (ViT S = 197), model/transformer.py:115-130 (AST S = 129)."""
    from valor_amd import kernels as K, lib
    so = lib.load()
    E = H * 64
    g = torch.Generator().manual_seed(S * 7 + B)
    qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
    dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
    scale = 1.0 / math.sqrt(64)
    o, lse = K.attn_fwd(q, k, v, H, scale=scale, p_drop=p_drop, seed=5, offset=9)
    old = so.valor_attn_set_res_pipeline(1)
    try:
        dq1, dk1, dv1 = K.attn_bwd(q, k, v, o, lse, dout, H, scale=scale, p_drop=p_drop, seed=5, offset=9)
        dq1b, dk1b, dv1b = K.attn_bwd(q, k, v, o, lse, dout, H, scale=scale, p_drop=p_drop, seed=5, offset=9)     # and it is deterministic
        so.valor_attn_set_res_pipeline(0)
        dq0, dk0, dv0 = K.attn_bwd(q, k, v, o, lse, dout, H, scale=scale, p_drop=p_drop, seed=5, offset=9)
        so.valor_attn_set_res_pipeline(2)      # 16 waves x 16-row blocks: every output row accumulates in the order of mode 0
        dq2, dk2, dv2 = K.attn_bwd(q, k, v, o, lse, dout, H, scale=scale, p_drop=p_drop, seed=5, offset=9)
        # mode 3: the FIRST version of the pipelined kernel (operands from global memory at the top of each phase, 8-byte stores); mode 1 is the
        # second (round 6: operands prefetched / out of LDS, 16-byte stores through v_permlane16_swap) -- the same arithmetic, bit for bit
        so.valor_attn_set_res_pipeline(3)
        dq3, dk3, dv3 = K.attn_bwd(q, k, v, o, lse, dout, H, scale=scale, p_drop=p_drop, seed=5, offset=9)
        so.valor_attn_set_res_pipeline(1)      # gradients written into one packed [B, S, 3E] buffer (the model's layout: row stride 3E)
        dqkv = torch.full_like(qkv, float("nan"))
        K.attn_bwd(q, k, v, o, lse, dout, H, dq=dqkv[:, :, :E], dk=dqkv[:, :, E:2 * E], dv=dqkv[:, :, 2 * E:], scale=scale, p_drop=p_drop, seed=5, offset=9)
    finally:
        so.valor_attn_set_res_pipeline(old)
    torch.cuda.synchronize()
    assert torch.equal(dq2, dq0) and torch.equal(dk2, dk0) and torch.equal(dv2, dv0)
    if S > 160:            # (shorter sequences: mode 1 is the 16-wave kernel, mode 3 the pipelined one -- another summation order of delta)
        assert torch.equal(dq3, dq1) and torch.equal(dk3, dk1) and torch.equal(dv3, dv1)
    assert torch.equal(dqkv, torch.cat((dq1, dk1, dv1), dim=-1))
    for a, b_, c, n in ((dq1, dq0, dq1b, "dq"), (dk1, dk0, dk1b, "dk"), (dv1, dv0, dv1b, "dv")):
        assert torch.equal(a, c), n
        assert _rel(a, b_) < 5e-4, (n, _rel(a, b_))
        assert float((a.float() - b_.float()).abs().max()) <= 2.0 ** -7 * float(b_.float().abs().max()), n
    if p_drop == 0.0 and S == 197:
        sl = slice(0, 3)          # fp64 reference on three samples
        qd, kd, vd = (t[sl].double().detach().requires_grad_(True) for t in (q, k, v))
        oref = _ref_attn(qd, kd, vd, H, None, None, 0, scale)
        (oref * dout[sl].double()).sum().backward()
        assert _rel(dq1[sl], qd.grad) < 2e-2 and _rel(dk1[sl], kd.grad) < 2e-2 and _rel(dv1[sl], vd.grad) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Skv", [(5, 12, 2, 41), (3, 12, 1, 64), (2, 8, 4, 65), (7, 12, 2, 128), (1, 1, 3, 1), (130, 12, 2, 33), (3, 12, 2, 256), (2, 12, 4, 200)])
def test_decode_step_kernel(dev, dtype, B, H, Sq, Skv):
    """attn_dec_fwd_kernel (one wave per (sequence, head): the cached decoding step, <= 4 query rows against <= 256 keys) against explicit
    fp64 softmax attention and against the tiled kernel it replaces for these shapes (variant bit 2 off); q / k / v are strided views of
    a fused projection and of a [R, L, 2E] slot buffer as valor_amd/decode.py passes them, the mask closes a ragged tail of slots."""
    from valor_amd import kernels as K, lib
    E = H * 64
    g = torch.Generator().manual_seed(B * 100 + Sq * 10 + Skv)
    qkv = (torch.randn((B, Sq, 3 * E), generator=g) * 0.7).to(dev, dtype)
    slots = (torch.randn((B, Skv, 2 * E), generator=g) * 0.7).to(dev, dtype)
    mask = torch.zeros((B, Sq, Skv))
    for b in range(B):
        for j in range(Sq):
            mask[b, j, 1 + (b * 7 + j * 3) % Skv:] = -10000.0
    mask = mask.to(dev)
    q, k, v = qkv[:, :, :E], slots[:, :, :E], slots[:, :, E:]
    so = lib.load()
    assert so.valor_attn_set_variant(-1) & 4
    o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=0.125)
    old = so.valor_attn_set_variant(3)
    try:
        o_t, lse_t = K.attn_fwd(q, k, v, H, mask=mask, scale=0.125)
    finally:
        so.valor_attn_set_variant(old)
    want = _ref_attn(q.double(), k.double(), v.double(), H, mask, None, 0, 0.125)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert torch.isfinite(o.float()).all()
    assert _rel(o, want) < tol, _rel(o, want)
    assert _rel(o, want) <= 1.5 * _rel(o_t, want) + 1e-6, (_rel(o, want), _rel(o_t, want))
    assert (lse - lse_t).abs().max().item() < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Skv", [(6, 12, 2, 41), (9, 8, 1, 130)])
def test_decode_step_kernel_reads_slots_through_a_row_table(dev, dtype, B, H, Sq, Skv):
    """valor_attn_decode_fwd with key_row (beam search: slot j of sequence b lives in batch row key_row[b, j]) equals the same kernel on
    K / V gathered into place, bit for bit, and a null table equals the identity table."""
    from valor_amd import kernels as K
    E = H * 64
    g = torch.Generator().manual_seed(B + Skv)
    qkv = (torch.randn((B, Sq, 3 * E), generator=g) * 0.7).to(dev, dtype)
    slots = (torch.randn((B, Skv, 2 * E), generator=g) * 0.7).to(dev, dtype)
    mask = torch.zeros((B, Sq, Skv))
    mask[:, :, Skv - 3:] = -10000.0
    mask = mask.to(dev)
    table = torch.randint(0, B, (B, Skv), generator=g).to(dev, torch.int32)
    q, k, v = qkv[:, :, :E], slots[:, :, :E], slots[:, :, E:]
    got = K.attn_decode(q, k, v, H, mask=mask, key_row=table)
    moved = torch.gather(slots, 0, table.long()[:, :, None].expand(B, Skv, 2 * E)).contiguous()
    want = K.attn_decode(q, moved[:, :, :E], moved[:, :, E:], H, mask=mask)
    assert torch.equal(got, want)
    ident = torch.arange(B, device=dev, dtype=torch.int32)[:, None].expand(B, Skv).contiguous()
    assert torch.equal(K.attn_decode(q, k, v, H, mask=mask, key_row=ident), K.attn_decode(q, k, v, H, mask=mask))
    o, _ = K.attn_fwd(q, moved[:, :, :E], moved[:, :, E:], H, mask=mask, scale=0.125)
    assert torch.equal(o, want)                       # valor_attn_fwd takes the same kernel for these shapes
