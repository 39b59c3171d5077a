"""GPU parity of every kernel FAMILY behind valor_gemm / valor_attn_* (the per-shape policy picks between them, so each
is tested explicitly through the variant switches): bf16 results against fp64 torch math on the layouts / tails / grouped
ranges / dropout windows each family has its own code for."""
import itertools
import math

import pytest
import torch

from test_attention_gpu import _ref_attn, _rel as _arel

pytestmark = pytest.mark.gpu


def _mk(shape, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / max(b.norm().item(), 1e-6)).item()


@pytest.fixture
def gemm_variant():
    from valor_amd import lib
    so = lib.load()
    old = so.valor_gemm_set_variant(-1)
    yield so
    so.valor_gemm_set_variant(old)


@pytest.fixture
def attn_variant():
    from valor_amd import lib
    so = lib.load()
    old = so.valor_attn_set_variant(-1)
    yield so
    so.valor_attn_set_variant(old)


GEMM_SHAPES = [(256, 256, 128), (512, 768, 64), (304, 520, 192), (1000, 264, 4160), (4104, 96, 40), (136, 2056, 1000), (8, 8, 8)]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_gemm_families(dev, gemm_variant, variant):
    from valor_amd import kernels as K, lib
    gemm_variant.valor_gemm_set_variant(variant)
    for (M, N, Kd), ta, tb in itertools.product(GEMM_SHAPES, [False, True], [False, True]):
        A = _mk((Kd, M) if ta else (M, Kd), 1, dev)
        B = _mk((Kd, N) if tb else (N, Kd), 2, dev)
        ref = (A.t() if ta else A).double() @ (B if tb else B.t()).double()
        for sk in (False, True):
            C = K.gemm(A, B, trans_a=ta, trans_b=tb, splitk=sk)
            assert _rel(C, ref) < 6e-3, (variant, M, N, Kd, ta, tb, sk, _rel(C, ref))
    # variants 0-3 pin one family; under the measured policy (4) a 6-tile forward problem belongs to the 256 x 128 kernel (family 4)
    fam = gemm_variant.valor_gemm_kernel_for(lib.DT_BF16, 0, 0, 512, 768, 128, 0)
    assert fam == variant
    # epilogues: bias + erf-GELU + saved pre-activation, dact multiply, fp32 accumulate
    X, W, b = _mk((512, 256), 6, dev), _mk((768, 256), 7, dev), _mk((768,), 8, dev)
    out, pre = K.gemm(X, W, bias=b, act=lib.ACT_GELU_ERF, want_preact=True, splitk=False)
    pr = X.double() @ W.double().t() + b.double()
    assert _rel(pre, pr) < 6e-3 and _rel(out, torch.nn.functional.gelu(pr)) < 6e-3
    W2, dY2 = _mk((128, 768), 9, dev), _mk((512, 128), 10, dev)
    dU = K.gemm(dY2, W2, trans_b=True, act=lib.ACT_GELU_ERF, dact_aux=pre, splitk=False)       # (dY2 . W2) * gelu'(u)
    u = pre.double()
    dact = 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
    assert _rel(dU, (dY2.double() @ W2.double()) * dact) < 8e-3
    Cacc = torch.ones((512, 768), dtype=torch.float32, device=dev)
    K.gemm(X, W, alpha=0.5, out=Cacc, accumulate=True, out_dtype=torch.float32, splitk=False)
    assert _rel(Cacc, 1.0 + 0.5 * (X.double() @ W.double().t())) < 6e-3
    # k tail with NaN-poisoned pads (direct operands) and a k-slow partner
    M, N, Kd, ldk = 70, 136, 1001, 1008
    Ab, Bb = _mk((M, ldk), 3, dev), _mk((N, ldk), 4, dev)
    Ab[:, Kd:] = float("nan"); Bb[:, Kd:] = float("nan")
    C = K.gemm(Ab[:, :Kd], Bb[:, :Kd], splitk=False)
    assert _rel(C, Ab[:, :Kd].double() @ Bb[:, :Kd].double().t()) < 6e-3
    Bt = _mk((Kd, 136), 5, dev)
    C2 = K.gemm(Ab[:, :Kd], Bt, trans_b=True, splitk=False)
    assert _rel(C2, Ab[:, :Kd].double() @ Bt.double()) < 6e-3


def test_gemm_8phase_transposing_reads_both_ways(dev, gemm_variant):
    """k-slow 8-phase kernels with the transposing LDS reads as compiler builtins (0) and as inline asm (1, the default: keeps the
    counted LDS-DMA pipeline from being drained): same results either way, fused row sums included"""
    from valor_amd import kernels as K, lib
    gemm_variant.valor_gemm_set_variant(3)
    old = gemm_variant.valor_gemm_set_tr_asm(-1)
    try:
        outs = []
        for mode in (0, 1):
            gemm_variant.valor_gemm_set_tr_asm(mode)
            res = []
            for (M, N, Kd), ta, tb in itertools.product([(768, 520, 4096), (304, 1000, 192)], [False, True], [False, True]):
                A = _mk((Kd, M) if ta else (M, Kd), 1, dev)
                B = _mk((Kd, N) if tb else (N, Kd), 2, dev)
                ref = (A.t() if ta else A).double() @ (B if tb else B.t()).double()
                for sk in (False, True):
                    C = K.gemm(A, B, trans_a=ta, trans_b=tb, splitk=sk)
                    assert _rel(C, ref) < 6e-3, (mode, M, N, Kd, ta, tb, sk)
                    res.append(C)
            dY, X = _mk((4096, 768), 11, dev), _mk((4096, 768), 12, dev)
            rs = torch.zeros(768, dtype=torch.bfloat16, device=dev)
            res.append(K.gemm(dY, X, trans_a=True, trans_b=True, rowsum_out=rs)); res.append(rs)
            outs.append(res)
        for a, b in zip(*outs):
            assert torch.equal(a, b)                    # the same arithmetic in the same order: bit identical
    finally:
        gemm_variant.valor_gemm_set_tr_asm(old)


def test_gemm_8phase_fast_epilogue(dev, gemm_variant):
    """one-pass bf16 epilogue of plain 8-phase problems (bias / activation in registers) against the general two-pass epilogue
    and fp64 math: every layout, row tails, bias + GELU, alpha; problems it does not cover fall back untouched"""
    from valor_amd import kernels as K, lib
    gemm_variant.valor_gemm_set_variant(3)
    old = gemm_variant.valor_gemm_set_fast_epilogue(-1)
    try:
        for (M, N, Kd), ta, tb in itertools.product([(1000, 520, 256), (512, 768, 4096), (304, 264, 192)], [False, True], [False, True]):
            A = _mk((Kd, M) if ta else (M, Kd), 1, dev)
            B = _mk((Kd, N) if tb else (N, Kd), 2, dev)
            bias = _mk((N,), 3, dev)
            ref = 0.5 * ((A.t() if ta else A).double() @ (B if tb else B.t()).double()) + bias.double()
            outs = []
            for mode in (0, 1):
                gemm_variant.valor_gemm_set_fast_epilogue(mode)
                outs.append(K.gemm(A, B, trans_a=ta, trans_b=tb, bias=bias, alpha=0.5, splitk=False))
                assert _rel(outs[-1], ref) < 6e-3, (mode, M, N, Kd, ta, tb)
            assert _rel(outs[1], outs[0]) < 1e-3
        # fused epilogues of the bf16 tile path against the general fp32 two-pass epilogue (mode 0) and fp64 math: bias + activation,
        # activation with the pre-activation copy (two tile passes over the same fp32 accumulators), the
        # act' multiply of a dgrad and C += (applied at read-out: one extra bf16 rounding of the GEMM result), row / column tails
        import torch.nn.functional as F
        for (M, N, Kd), tb in itertools.product([(512, 768, 256), (1000, 520, 192)], [False, True]):
            X, W, b = _mk((M, Kd), 6, dev), _mk((Kd, N) if tb else (N, Kd), 7, dev), _mk((N,), 8, dev)
            U = _mk((M, N), 9, dev)
            z = X.double() @ (W.double() if tb else W.double().t()) + b.double()
            res = {}
            for mode in (0, 2):           # 2 = the tile path wherever it is implemented (1, the default, leaves preact / dact on the general one)
                gemm_variant.valor_gemm_set_fast_epilogue(mode)
                y = K.gemm(X, W, trans_b=tb, bias=b, act=lib.ACT_GELU_ERF, splitk=False)
                out, pre = K.gemm(X, W, trans_b=tb, bias=b, act=lib.ACT_QUICK_GELU, want_preact=True, splitk=False)
                du = K.gemm(X, W, trans_b=tb, act=lib.ACT_GELU_ERF, dact_aux=U, splitk=False)
                Cacc = torch.ones((M, N), dtype=torch.bfloat16, device=dev)
                K.gemm(X, W, trans_b=tb, out=Cacc, accumulate=True, splitk=False)
                # derivative-saving pair (ACT_DERIV): act'(z) in the `preact` slot (always the general epilogue), a plain multiply in the dgrad
                # (the tile path in modes 1 and 2)
                out_d, gp = K.gemm(X, W, trans_b=tb, bias=b, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True, splitk=False)
                sg = torch.sigmoid(1.702 * z)
                assert _rel(out_d, z * sg) < 6e-3 and _rel(gp, sg + 1.702 * z * sg * (1 - sg)) < 6e-3
                du_d = K.gemm(X, W, trans_b=tb, act=lib.ACT_GELU_ERF | lib.ACT_DERIV, dact_aux=U, splitk=False)
                assert _rel(du_d, (z - b.double()) * U.double()) < 8e-3
                res[mode] = (y, out, pre, du, Cacc)
                assert _rel(y, F.gelu(z)) < 6e-3
                assert _rel(pre, z) < 6e-3 and _rel(out, z * torch.sigmoid(1.702 * z)) < 6e-3
                ud = U.double().requires_grad_(True)
                (g,) = torch.autograd.grad(F.gelu(ud).sum(), ud)
                assert _rel(du, (z - b.double()) * g) < 8e-3
                assert _rel(Cacc, 1.0 + z - b.double()) < 6e-3
            assert max(_rel(res[2][i], res[0][i]) for i in range(3)) < 1e-3          # same fp32 accumulators, rounded once in both
            assert _rel(res[2][3], res[0][3]) < 6e-3 and _rel(res[2][4], res[0][4]) < 6e-3
    finally:
        gemm_variant.valor_gemm_set_fast_epilogue(old)


def test_gemm_fused_rowsum(dev, gemm_variant):
    """bias gradient beside the wgrad GEMM (valor_gemm's rowsum_out): sum over tokens of dY computed on the matrix pipe in the
    8-phase k-slow kernel, with and without split-K, plain and accumulating, fp32 and bf16 outputs; refused elsewhere."""
    from valor_amd import kernels as K, lib
    gemm_variant.valor_gemm_set_variant(3)
    for (M, N, Kd) in [(768, 768, 4096), (3072, 768, 12608), (520, 264, 192), (256, 3072, 1024)]:
        dY, X = _mk((Kd, M), 11, dev), _mk((Kd, N), 12, dev)
        assert K.gemm_fuses_rowsum(dY, X, True, True)
        ref = dY.double().sum(0)
        for sk, odt in itertools.product((False, True), (torch.float32, torch.bfloat16)):
            rs = torch.full((M,), 2.0, dtype=odt, device=dev)
            C = K.gemm(dY, X, trans_a=True, trans_b=True, splitk=sk, out_dtype=odt, rowsum_out=rs, rowsum_accumulate=True)
            assert _rel(C, dY.double().t() @ X.double()) < 6e-3
            tol = 2e-5 if odt == torch.float32 else 6e-3
            assert _rel(rs, ref + 2.0) < tol, (M, N, Kd, sk, odt, _rel(rs, ref + 2.0))
            rs2 = torch.full((M,), float("nan"), dtype=odt, device=dev)
            K.gemm(dY, X, trans_a=True, trans_b=True, splitk=sk, out_dtype=odt, rowsum_out=rs2)
            assert _rel(rs2, ref) < tol
    gemm_variant.valor_gemm_set_variant(1)
    dY, X = _mk((4096, 768), 11, dev), _mk((4096, 768), 12, dev)
    assert not K.gemm_fuses_rowsum(dY, X, True, True)
    with pytest.raises(lib.ValorHipError):
        K.gemm(dY, X, trans_a=True, trans_b=True, rowsum_out=torch.zeros(768, dtype=torch.bfloat16, device=dev))


@pytest.mark.parametrize("variant", [0, 3])
def test_attention_families_self(dev, attn_variant, variant):
    from valor_amd import kernels as K
    attn_variant.valor_attn_set_variant(variant)
    scale = 1.0 / math.sqrt(64)
    for (B, H, S, masked) in [(2, 12, 197, False), (3, 12, 129, False), (4, 12, 32, True), (2, 12, 42, True), (2, 3, 256, False), (2, 2, 17, True)]:
        g = torch.Generator().manual_seed(100 + S)
        E = H * 64
        qkv = (torch.randn((B, S, 3 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
        q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
        mask = None
        if masked:
            lens = torch.randint(3, S + 1, (B,), generator=g)
            m = (torch.arange(S)[None, :] < lens[:, None]).float()[:, None, :].expand(B, S, S).clone()
            mask = ((1.0 - torch.tril(m)) * -10000.0).to(dev).contiguous()
        dout = torch.randn((B, S, E), generator=g).to(torch.bfloat16).to(dev)
        qd, kd, vd = (t.double().detach().requires_grad_(True) for t in (q, k, v))
        oref = _ref_attn(qd, kd, vd, H, mask, None, 0, scale)
        (oref * dout.double()).sum().backward()
        o, lse = K.attn_fwd(q, k, v, H, mask=mask, scale=scale)
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, dout, H, mask=mask, scale=scale)
        assert _arel(o, oref) < 1e-2 and _arel(dq, qd.grad) < 2e-2 and _arel(dk, kd.grad) < 2e-2 and _arel(dv, vd.grad) < 2e-2, (variant, S)


@pytest.mark.parametrize("variant", [1, 3])
def test_attention_families_cross(dev, attn_variant, variant):
    from valor_amd import kernels as K
    attn_variant.valor_attn_set_variant(variant)
    scale = 1.0 / math.sqrt(64)
    cases = [(2, 12, 32, 458, 0, None), (6, 3, 32, 1834, 2, [(0, 1834), (0, 1576), (1576, 258)]), (2, 3, 42, 1834, 2, [(0, 1834)]),
             (4, 2, 48, 700, 2, [(0, 700), (130, 333)]), (3, 2, 17, 129, 3, [(5, 100)]), (8, 2, 16, 300, 2, [(0, 300), (0, 10), (290, 10), (100, 100)])]
    for (B, H, Sq, Skv, bmod, ranges) in cases:
        g = torch.Generator().manual_seed(5 + Sq + Skv)
        E = H * 64
        Bkv = bmod if bmod > 0 else B
        q = (torch.randn((B, Sq, 2 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)[:, :, :E]
        kvb = (torch.randn((Bkv, Skv, 2 * E), generator=g) * 0.8).to(torch.bfloat16).to(dev)
        k, v = kvb[:, :, :E], kvb[:, :, E:]
        kvr = torch.tensor([list(ranges[b // bmod]) for b in range(B)], dtype=torch.int32) if ranges is not None else None
        dout = torch.randn((B, Sq, E), generator=g).to(torch.bfloat16).to(dev)
        qd, kd, vd = (t.double().detach().requires_grad_(True) for t in (q, k, v))
        oref = _ref_attn(qd, kd, vd, H, None, kvr, bmod, scale)
        (oref * dout.double()).sum().backward()
        kvr_d = kvr.to(dev) if kvr is not None else None
        o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr_d, kv_bmod=bmod, scale=scale)
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, dout, H, kv_range=kvr_d, kv_bmod=bmod, scale=scale)
        assert _arel(o, oref) < 1e-2 and _arel(dq, qd.grad) < 2e-2 and _arel(dk, kd.grad) < 2e-2 and _arel(dv, vd.grad) < 2e-2, (variant, Sq, Skv)


def test_attention_dropout_same_mask_in_every_family(dev, attn_variant):
    """bf16 dropout: keep fraction, forward/backward agreement inside a family, and identical masks ACROSS families
    (the per-element hash is layout free) -- self-attention and grouped cross-attention."""
    from valor_amd import kernels as K
    pd = 0.25
    # self attention, S = 128
    B, H, S = 2, 2, 128
    E = H * 64
    q = torch.zeros((B, S, E), device=dev, dtype=torch.bfloat16)
    k = torch.randn((B, S, E), device=dev).to(torch.bfloat16)
    v = torch.eye(S, device=dev)[:, :64].repeat(1, H)[None].expand(B, S, E).contiguous().to(torch.bfloat16)
    keeps = []
    for var in (0, 3):
        attn_variant.valor_attn_set_variant(var)
        o, lse = K.attn_fwd(q, k, v, H, p_drop=pd, seed=7, offset=11)
        keep = (o.float() * S * (1 - pd) > 0.5)
        assert abs(keep.float().mean().item() - (1 - pd)) < 0.02
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, torch.ones_like(o), H, p_drop=pd, seed=7, offset=11)
        want = keep.float().view(B, S, H, 64).sum(1) / (1 - pd) / S
        got = dv.float().view(B, S, H, 64)[:, :64, :, 0].permute(0, 2, 1)
        assert torch.allclose(got, want, atol=2e-2), (var, (got - want).abs().max())
        keeps.append(keep)
    assert torch.equal(keeps[0], keeps[1])
    # grouped cross attention
    B, H, Sq, Skv, bmod = 4, 2, 32, 256, 2
    E = H * 64
    q = torch.zeros((B, Sq, E), device=dev, dtype=torch.bfloat16)
    k = torch.randn((bmod, Skv, E), device=dev).to(torch.bfloat16)
    v = torch.eye(Skv, device=dev)[:, :64].repeat(1, H)[None].expand(bmod, Skv, E).contiguous().to(torch.bfloat16)
    kvr = torch.tensor([[0, 256], [0, 256], [0, 128], [0, 128]], dtype=torch.int32).to(dev)
    n = torch.tensor([256, 256, 128, 128], device=dev).view(B, 1, 1).float()
    keeps = []
    for var in (1, 3):
        attn_variant.valor_attn_set_variant(var)
        o, lse = K.attn_fwd(q, k, v, H, kv_range=kvr, kv_bmod=bmod, p_drop=pd, seed=7, offset=11)
        keep = (o.float() * n * (1 - pd) > 0.5)
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, torch.ones_like(o), H, kv_range=kvr, kv_bmod=bmod, p_drop=pd, seed=7, offset=11)
        want = (keep.float() / (1 - pd) / n).view(B, Sq, H, 64).sum(1).view(B // bmod, bmod, H, 64).sum(0)
        got = dv.float().view(bmod, Skv, H, 64)[:, :64, :, 0].permute(0, 2, 1)
        assert torch.allclose(got, want, atol=2e-2), (var, (got - want).abs().max())
        keeps.append(keep)
    assert torch.equal(keeps[0], keeps[1])


@pytest.mark.parametrize("cols,rows,p,scaled", [(768, 1001, 0.0, False), (768, 514, 0.1, False), (1024, 333, 0.1, True), (512, 77, 0.0, False), (256, 9, 0.25, True),
                                                (128, 4099, 0.0, False), (128, 61, 0.2, True), (192, 1003, 0.1, True), (384, 258, 0.0, False)])
def test_layernorm_families_agree(dev, cols, rows, p, scaled):
    """fused bias + dropout + residual + LayerNorm: one-wave-per-row kernels (variant 0) vs the kernels with 16-byte accesses and
    several rows per wave -- half a wave per row at 256 / 512 / 768 / 1024 columns, a quarter at 128 / 384, an eighth at 192 (the
    VideoSwin stage-1 widths, videoswin.py:191-245) -- (variant 2 = forward AND backward): bit-identical z (same Philox windows, same adds),
    y / statistics / dx / dres and the column sums dgamma / dbeta / dbias equal up to the reduction order; odd row counts leave lane
    groups without a row."""
    from valor_amd import kernels as K, lib
    so = lib.load()
    old = so.valor_ln_set_variant(-1)
    g = torch.Generator().manual_seed(cols + rows)
    mk = lambda *s: torch.randn(s, generator=g).to(torch.bfloat16).to(dev)
    x, res, bias, gam, bet, dy, dz = mk(rows, cols), mk(rows, cols), mk(cols), mk(cols), mk(cols), mk(rows, cols), mk(rows, cols)
    rs = (torch.rand((rows + 6) // 7, generator=g) > 0.3).float().mul(1.25).to(dev) if scaled else None
    outs = {}
    try:
        for v in (0, 2):
            so.valor_ln_set_variant(v)
            z, y, mean, rstd = K.bdrln_fwd(x, bias, res, gam, bet, 1e-5, p_drop=p, seed=5, offset=3, row_scale=rs, rows_per_scale=7 if scaled else 0)
            dx, dres, dg, db, dbias = K.bdrln_bwd(dy, dz, z, mean, rstd, gam, p_drop=p, seed=5, offset=3, want_dbias=True, row_scale=rs,
                                                  rows_per_scale=7 if scaled else 0)
            outs[v] = [t.float().clone() for t in (z, y, mean, rstd, dx, dres, dg, db, dbias)]
    finally:
        so.valor_ln_set_variant(old)
    assert torch.equal(outs[0][0], outs[2][0])
    for name, a, b in zip(("y", "mean", "rstd", "dx", "dres", "dgamma", "dbeta", "dbias"), outs[2][1:], outs[0][1:]):
        assert float((a - b).norm()) <= 4e-3 * float(b.norm()) + 1e-6, name


@pytest.mark.parametrize("rows,p,with_dz", [(1001, 0.0, True), (514, 0.1, True), (70001, 0.1, True), (4097, 0.0, False)])
def test_layernorm_backward_with_lds_column_accumulators(dev, rows, p, with_dz):
    """the half-wave backward with its three column accumulators in a per-wave LDS array (csrc/layernorm.hip LACC; variant 3 forces it, the
    default picks it for >= 49 152 rows of 768 columns with a residual gradient coming in -- the 70 001-row case runs it under the DEFAULT
    variant): dx / dres bit-identical to the register-accumulator half-wave kernel (the same row arithmetic), column sums equal up to the
    reduction order; odd row counts leave a half-wave without a row."""
    from valor_amd import kernels as K, lib
    so = lib.load()
    old = so.valor_ln_set_variant(-1)
    cols = 768
    g = torch.Generator().manual_seed(rows)
    mk = lambda *s: torch.randn(s, generator=g).to(torch.bfloat16).to(dev)
    x, res, bias, gam, bet, dy = mk(rows, cols), mk(rows, cols), mk(cols), mk(cols), mk(cols), mk(rows, cols)
    dz = mk(rows, cols) if with_dz else None
    outs = {}
    try:
        so.valor_ln_set_variant(1)
        z, y, mean, rstd = K.bdrln_fwd(x, bias, res, gam, bet, 1e-5, p_drop=p, seed=5, offset=3)
        for v in (1, 2, 3):
            so.valor_ln_set_variant(v)
            dx, dres, dg, db, dbias = K.bdrln_bwd(dy, dz, z, mean, rstd, gam, p_drop=p, seed=5, offset=3, want_dbias=True)
            outs[v] = [t.float().clone() for t in (dx, dres, dg, db, dbias)]
    finally:
        so.valor_ln_set_variant(old)
    assert torch.equal(outs[2][0], outs[3][0]) and torch.equal(outs[2][1], outs[3][1])
    if rows >= 49152 and with_dz:                     # the default rule took the LDS-accumulator kernel
        assert torch.equal(outs[1][0], outs[3][0])
    for v in (1, 2):
        for name, a, b in zip(("dx", "dres", "dgamma", "dbeta", "dbias"), outs[3], outs[v]):
            assert float((a - b).norm()) <= 4e-3 * float(b.norm()) + 1e-6, (v, name)


def test_dropout_is_unbiased_in_expectation(dev):
    """E[dropout(x) / (1 - p)] = x for both counter-based generators: the attention probability hash (attn_common.h) and the
    Philox4x32-10 windows of the fused LayerNorm kernels. The mean over N independent windows must approach the p = 0 output at the
    binomial rate: |mean - exact| ~ sqrt(p / ((1 - p) N)) * |contribution|, checked at 5 sigma on aggregate norms."""
    from valor_amd import kernels as K
    p, N = 0.1, 96
    # ---- attention: O = sum_k dropout(P)_k V_k
    B, H, S = 2, 4, 197
    E = H * 64
    g = torch.Generator().manual_seed(4)
    q, k, v = ((torch.randn((B, S, E), generator=g) * 0.7).to(torch.bfloat16).to(dev) for _ in range(3))
    o0, _ = K.attn_fwd(q, k, v, H, p_drop=0.0)
    acc = torch.zeros_like(o0, dtype=torch.float32)
    for i in range(N):
        o, _ = K.attn_fwd(q, k, v, H, p_drop=p, seed=11, offset=1000003 * i)
        acc += o.float()
    err = float((acc / N - o0.float()).norm() / o0.float().norm())
    # one draw deviates by ~ sqrt(p/(1-p)) * sqrt(sum P^2 V^2) / |O| ~ 0.33 * O(1) for near-uniform P; N draws: / sqrt(N)
    one, _ = K.attn_fwd(q, k, v, H, p_drop=p, seed=11, offset=5)
    single = float((one.float() - o0.float()).norm() / o0.float().norm())
    assert 0.05 < single < 1.0
    assert err < 2.0 * single / N ** 0.5 + 4e-3, (err, single)
    # ---- fused bias + dropout + residual (+ LayerNorm): z = dropout(x + b) / (1 - p) + res
    rows, cols = 512, 768
    x, res = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev), torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev)
    bias = torch.randn((cols,), generator=g).to(torch.bfloat16).to(dev)
    exact = x.float() + bias.float() + res.float()
    acc = torch.zeros((rows, cols), dtype=torch.float32, device=dev)
    keep = 0.0
    for i in range(N):
        z, _, _, _ = K.bdrln_fwd(x, bias, res, None, None, 0.0, p_drop=p, seed=3, offset=7919 * i, want_y=False)
        acc += z.float()
        keep += float(((z.float() - res.float()).abs() > 0).float().mean())
    assert abs(keep / N - (1 - p)) < 2e-3
    err = float((acc / N - exact).norm() / (x.float() + bias.float()).norm())
    assert err < 2.0 * (p / (1 - p) / N) ** 0.5 + 4e-3, err
