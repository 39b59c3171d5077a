"""GPU parity of the GEMM / fused-LN / colsum HIP kernels against plain torch fp32/fp64 math."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    # relative L2 error; a reference that is (numerically) zero falls back to an absolute scale of 1e-3/elt
    a, b = a.double(), b.double()
    return ((a - b).norm() / max(b.norm().item(), 5e-2 * b.numel() ** 0.5)).item()


def _mk(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


SHAPES = [(128, 128, 64), (256, 384, 128), (197, 130, 72), (1, 5, 8), (300, 30522 // 16, 768), (4100, 96, 40)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", list(itertools.product([False, True], [False, True])))
def test_gemm_layouts(dev, dtype, ta, tb):
    from valor_amd import kernels as K
    vec = 8 if dtype == torch.bfloat16 else 4
    for (M, N, Kd) in SHAPES:
        if ta and M % vec: M += vec - M % vec        # transposed operands need 16-B aligned rows
        if tb and N % vec: N += vec - N % vec
        if not ta and Kd % vec: Kd += vec - Kd % vec
        if not tb and Kd % vec: Kd += vec - Kd % vec
        A = _mk((Kd, M) if ta else (M, Kd), dtype, dev, 1)
        B = _mk((Kd, N) if tb else (N, Kd), dtype, dev, 2)     # asymmetric random B (transpose-detecting)
        C = K.gemm(A, B, trans_a=ta, trans_b=tb, splitk=False)
        Af = (A.t() if ta else A).double()
        Bf = (B.t() if tb else B).double()
        ref = Af @ Bf.t()
        tol = 2e-6 if dtype == torch.float32 else 6e-3
        assert C.shape == ref.shape
        assert _rel(C, ref) < tol, (M, N, Kd, _rel(C, ref))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_ktail_and_padded_ld(dev, dtype):
    """K not a multiple of the chunk (vocab 30522-style) with a zero-padded leading dim."""
    from valor_amd import kernels as K
    M, N, Kd, ldk = 70, 136, 1001, 1008
    Abuf = _mk((M, ldk), dtype, dev, 3); Bbuf = _mk((N, ldk), dtype, dev, 4)
    Abuf[:, Kd:] = float("nan"); Bbuf[:, Kd:] = float("nan")      # pads must never be read into the result
    A, B = Abuf[:, :Kd], Bbuf[:, :Kd]
    C = K.gemm(A, B, splitk=False)
    ref = A.double() @ B.double().t()
    assert _rel(C, ref) < (2e-6 if dtype == torch.float32 else 6e-3)
    # transposed B with contraction bound (rows >= K zero-filled)
    Bt = _mk((Kd, 136), dtype, dev, 5)
    C2 = K.gemm(A, Bt, trans_b=True, splitk=False)
    ref2 = A.double() @ Bt.double()
    assert _rel(C2, ref2) < (2e-6 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogues(dev, dtype, act):
    from valor_amd import kernels as K
    M, N, Kd = 200, 264, 96
    A = _mk((M, Kd), dtype, dev, 6, 0.5); W = _mk((N, Kd), dtype, dev, 7, 0.2); b = _mk((N,), dtype, dev, 8)
    out, pre = K.gemm(A, W, bias=b, act=act, want_preact=True, splitk=False)
    u = A.double() @ W.double().t() + b.double()
    f = {0: lambda x: x, 1: lambda x: x * 0.5 * (1 + torch.erf(x / 2 ** 0.5)),
         2: lambda x: x * torch.sigmoid(1.702 * x), 3: torch.relu}[act]
    tol = 3e-6 if dtype == torch.float32 else 8e-3
    assert _rel(pre, u) < tol
    assert _rel(out, f(u)) < tol
    # dact epilogue: G = (dY @ W2) * act'(u)
    dY = _mk((M, 64), dtype, dev, 9); W2 = _mk((64, N), dtype, dev, 10, 0.3)
    G = K.gemm(dY, W2, trans_b=True, act=act, dact_aux=pre, splitk=False)
    ud = pre.double().requires_grad_(True)
    f(ud).backward((dY.double() @ W2.double()))
    assert _rel(G, ud.grad) < (5e-6 if dtype == torch.float32 else 1e-2)
    # derivative-saving pair (lib.ACT_DERIV, what ops.MlpFn runs): `preact` receives act'(u), the dgrad multiplies by it
    from valor_amd import lib
    out_d, gprime = K.gemm(A, W, bias=b, act=act | lib.ACT_DERIV, want_preact=True, splitk=False)
    ud2 = u.clone().requires_grad_(True)
    (gp_ref,) = torch.autograd.grad(f(ud2).sum(), ud2)
    assert _rel(out_d, f(u)) < tol
    assert _rel(gprime, gp_ref) < (5e-6 if dtype == torch.float32 else 8e-3)
    G2 = K.gemm(dY, W2, trans_b=True, act=act | lib.ACT_DERIV, dact_aux=gprime, splitk=False)
    assert _rel(G2, (dY.double() @ W2.double()) * gprime.double()) < (5e-6 if dtype == torch.float32 else 8e-3)
    assert _rel(G2, (dY.double() @ W2.double()) * gp_ref) < (5e-6 if dtype == torch.float32 else 1.2e-2)
    # accumulate + alpha + fp32 output
    Cacc = torch.ones((M, N), dtype=torch.float32, device=dev)
    K.gemm(A, W, alpha=0.5, out=Cacc, accumulate=True, out_dtype=torch.float32, splitk=False)
    assert _rel(Cacc, 1.0 + 0.5 * (A.double() @ W.double().t())) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_wgrad(dev, dtype):
    """weight-gradient shape: small output, long contraction -> split-K path."""
    from valor_amd import kernels as K
    Mtok, N, Kd = 8192 + 40, 256, 192
    dY = _mk((Mtok, N), dtype, dev, 11, 0.1); X = _mk((Mtok, Kd), dtype, dev, 12)
    dW = K.gemm(dY, X, trans_a=True, trans_b=True, splitk=True)
    ref = dY.double().t() @ X.double()
    assert _rel(dW, ref) < (3e-6 if dtype == torch.float32 else 6e-3)
    dW2 = K.gemm(dY, X, trans_a=True, trans_b=True, splitk=False)
    assert _rel(dW2, ref) < (3e-6 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols", [768, 512, 1024, 128, 1536])
def test_bdrln_fwd_bwd(dev, dtype, cols):
    from valor_amd import kernels as K
    rows = 333
    x = _mk((rows, cols), dtype, dev, 1); bias = _mk((cols,), dtype, dev, 2); res = _mk((rows, cols), dtype, dev, 3)
    g = _mk((cols,), dtype, dev, 4) + 1.0; be = _mk((cols,), dtype, dev, 5)
    dy = _mk((rows, cols), dtype, dev, 6); dzin = _mk((rows, cols), dtype, dev, 7)
    eps = 1e-12 if cols == 768 else 1e-5
    z, y, mean, rstd = K.bdrln_fwd(x, bias, res, g, be, eps)
    xd, bd, rd, gd, bed = [t.double().requires_grad_(True) for t in (x, bias, res, g, be)]
    zr = xd + bd + rd
    zr.retain_grad()
    yr = torch.nn.functional.layer_norm(zr, (cols,), gd, bed, eps)
    tol = 3e-6 if dtype == torch.float32 else 8e-3
    assert _rel(z, zr) < tol and _rel(y, yr) < tol
    if dtype == torch.float32:
        assert _rel(mean, zr.mean(-1)) < 1e-5
    (yr * dy.double()).sum().backward(retain_graph=True)
    (zr * dzin.double()).sum().backward()
    # feed the fp64 z for fp32 runs so backward is compared like-for-like
    dx, dres, dg, db, dbias = K.bdrln_bwd(dy, dzin, z, mean, rstd, g, want_dbias=True)
    tolb = 2e-5 if dtype == torch.float32 else 2e-2
    assert dx.data_ptr() == dres.data_ptr()
    assert _rel(dres, xd.grad) < tolb
    assert _rel(dg, gd.grad) < tolb and _rel(db, bed.grad) < tolb and _rel(dbias, bd.grad) < tolb


def test_bdrln_dropout_consistency(dev):
    """dropout: mask statistics, scaling, and backward regenerates the same mask."""
    from valor_amd import kernels as K
    rows, cols, p = 512, 768, 0.1
    x = torch.ones((rows, cols), device=dev); res = torch.zeros_like(x)
    z, y, mean, rstd = K.bdrln_fwd(x, None, res, None, None, 1e-12, p_drop=p, seed=1234, offset=77)
    kept = (z != 0)
    assert abs(kept.float().mean().item() - (1 - p)) < 5e-3
    assert torch.allclose(z[kept], torch.full_like(z[kept], 1 / (1 - p)))
    z2, *_ = K.bdrln_fwd(x, None, res, None, None, 1e-12, p_drop=p, seed=1234, offset=77)
    assert torch.equal(z, z2)
    z3, *_ = K.bdrln_fwd(x, None, res, None, None, 1e-12, p_drop=p, seed=1235, offset=77)
    assert not torch.equal(z, z3)
    dz = torch.ones_like(x)
    dx, dres, *_ = K.bdrln_bwd(None, dz, None, None, None, None, p_drop=p, seed=1234, offset=77,
                               want_dgamma=False, want_dbeta=False)
    assert torch.equal(dx != 0, kept) and torch.allclose(dres, dz)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(dev, dtype):
    from valor_amd import kernels as K
    x = _mk((5000, 3072 + 8), dtype, dev, 3)[:, :3070]
    s = K.colsum(x)
    assert _rel(s, x.double().sum(0)) < (1e-5 if dtype == torch.float32 else 5e-3)


def test_gemm_operands_beyond_2gib(dev):
    """operands are staged through buffer descriptors with 32-bit offsets: valor_gemm cuts an operand of 2 GiB or more into
    launches (rows of a row-major A; the token contraction of a wgrad, accumulating) -- VideoSwin-L's stage-1 activations at
    b = 64 are 2.5 GB. Checked on row / column slices against fp32 torch matmul of the same bf16 data."""
    from valor_amd import kernels as K
    rows, Kd, N = 1_500_016, 768, 64                      # 2.3 GB of bf16
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((rows, Kd), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    assert A.numel() * 2 > (1 << 31)
    W = _mk((N, Kd), torch.bfloat16, dev, 2)
    bias = _mk((N,), torch.bfloat16, dev, 3)
    C = K.gemm(A, W, bias=bias)
    for sl in (slice(0, 300), slice(rows // 2 - 77, rows // 2 + 200), slice(rows - 300, rows)):
        ref = A[sl].float() @ W.float().t() + bias.float()
        assert _rel(C[sl], ref) < 6e-3
    # wgrad layout: dW[M, N] = dY^T . X over 1.5 M tokens, dY being the big operand; then accumulate on top
    X = _mk((rows, N), torch.bfloat16, dev, 4)
    M = 512
    dW = K.gemm(A[:, :M], X, trans_a=True, trans_b=True, out_dtype=torch.float32)        # a column view: ld = 768
    ref = torch.zeros((M, N), dtype=torch.float64, device=dev)
    for r0 in range(0, rows, 250_000):
        ref += A[r0:r0 + 250_000, :M].double().t() @ X[r0:r0 + 250_000].double()
    assert _rel(dW, ref) < 2e-3
    K.gemm(A[:, :M], X, trans_a=True, trans_b=True, out=dW, out_dtype=torch.float32, accumulate=True)
    assert _rel(dW, 2 * ref) < 2e-3
