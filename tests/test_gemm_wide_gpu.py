"""Family 4 on eight waves (csrc/gemm8w.hip, policy key 10): the 256 x 128 tile on 512-thread workgroups of 64 x 64 wave tiles, four waves
per SIMD. Forced onto every eligible NN problem (policy keys 8 = 1, 10 = 1) and compared with fp32 torch.matmul per 256 x 128 tile, and bit
for bit with the 32 x 32 x 16 instantiation of gemm8n.hip (same MFMA, same k order per accumulator): one / two / many K-tiles, tails in M
and N, the bf16 tile epilogue (plain, bias, activation, pre-activation copy, C +=), the general fp32 epilogue (derivative copy, fp32 out,
odd N), the bench's decoder / AST / ViT forward shapes, and a race screen of the counted-vmcnt ring.
Reference semantics: nn.Linear forward, model/bert.py:233-235,403-417, model/transformer.py:109-142, model/clip.py:176-192."""
import pytest
import torch

pytestmark = pytest.mark.gpu

W, I = 768, 3072
TOL, TILE_TOL = 2.5e-3, 4e-3


def _mk(shape, seed, dev, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def _tile_errors(C, ref, tm=256, tn=128):
    d = (C.float() - ref)
    M, N = ref.shape
    Mp, Np = (M + tm - 1) // tm * tm, (N + tn - 1) // tn * tn
    pad = lambda x: torch.nn.functional.pad(x, (0, Np - N, 0, Mp - M))
    e2 = pad(d * d).view(Mp // tm, tm, Np // tn, tn).sum(dim=(1, 3))
    r2 = pad(ref * ref).view(Mp // tm, tm, Np // tn, tn).sum(dim=(1, 3))
    return float(torch.sqrt(e2.sum() / r2.sum())), float(torch.sqrt(e2 / r2.clamp_min(1e-20)).max())


@pytest.fixture
def wide(dev):
    from valor_amd import lib
    so = lib.load()
    old8, old10 = so.valor_gemm_set_policy(8, 1), so.valor_gemm_set_policy(10, 1)
    yield so
    so.valor_gemm_set_policy(8, old8)
    so.valor_gemm_set_policy(10, old10)


def _m32(so, fn):
    """the same call on gemm8n.hip's 32 x 32 x 16 main loop (policy key 9) instead of the eight-wave kernel"""
    o10, o9 = so.valor_gemm_set_policy(10, 0), so.valor_gemm_set_policy(9, 1)
    try:
        return fn()
    finally:
        so.valor_gemm_set_policy(10, o10)
        so.valor_gemm_set_policy(9, o9)


def test_two_workgroups_per_cu(dev):
    """80 KiB of LDS and <= 128 registers per lane: two 512-thread workgroups = four waves per SIMD, the point of the kernel"""
    from valor_amd import lib
    assert lib.load().valor_gemm_wide_occupancy() == 2


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (256, 128, 64 * 3), (700, 328, 256), (1030, 1000, 448), (8832, 3 * W, W), (16512, W, I),
                                   (8832, 30522, W), (25216, I, W)])
def test_forward_nn(wide, dev, M, N, K):
    from valor_amd import kernels as Kn
    assert wide.valor_gemm_kernel_for(0, 0, 0, M, N, K, 0) == 4
    A, B, bias = _mk((M, K), 1, dev), _mk((N, K), 2, dev, 0.05), _mk((N,), 3, dev)
    ref = A.float() @ B.float().t()
    C = Kn.gemm(A, B)
    whole, worst = _tile_errors(C, ref)
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    assert torch.equal(C, _m32(wide, lambda: Kn.gemm(A, B)))
    Cb = Kn.gemm(A, B, bias=bias)
    whole, worst = _tile_errors(Cb, ref + bias.float())
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    assert torch.equal(Cb, _m32(wide, lambda: Kn.gemm(A, B, bias=bias)))
    c32 = Kn.gemm(A, B, bias=bias, out_dtype=torch.float32)
    assert float((c32 - ref - bias.float()).norm() / ref.norm()) < 2e-5


def test_fused_activation_epilogues_and_accumulate(wide, dev):
    """fc1 as ops.MlpFn launches it (bias + QuickGELU / erf-GELU, second output = act'(u) or u), with tails; C += into an existing buffer"""
    from valor_amd import kernels as Kn, lib
    for (M, N, K, s) in ((8832, I, W, 6), (700, 328, W, 61)):
        A, B, bias = _mk((M, K), s, dev), _mk((N, K), s + 1, dev, 0.05), _mk((N,), s + 2, dev, 0.5)
        u = A.float() @ B.float().t() + bias.float()
        sg = torch.sigmoid(1.702 * u)
        h, d = Kn.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True)
        for got, want in ((h, u * sg), (d, sg * (1 + 1.702 * u * (1 - sg)))):
            whole, worst = _tile_errors(got, want)
            assert whole < TOL and worst < TILE_TOL, (whole, worst)
        h2, u2 = Kn.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU, want_preact=True)
        assert torch.equal(h2, h)
        whole, worst = _tile_errors(u2, u)
        assert whole < TOL and worst < TILE_TOL, (whole, worst)
        g = Kn.gemm(A, B, bias=bias, act=lib.ACT_GELU_ERF)
        whole, worst = _tile_errors(g, torch.nn.functional.gelu(u))
        assert whole < TOL and worst < TILE_TOL, (whole, worst)
        hm, dm = _m32(wide, lambda: Kn.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True))
        assert torch.equal(h, hm) and torch.equal(d, dm)
        acc0 = _mk((M, N), s + 3, dev)
        out = acc0.clone()
        Kn.gemm(A, B, out=out, accumulate=True)
        whole, worst = _tile_errors(out, u - bias.float() + acc0.float())
        assert whole < 1.5 * TOL and worst < 1.5 * TILE_TOL, (whole, worst)


def test_repeated_launches_are_deterministic(wide, dev):
    """race screen of the ring (RAW: read after the barrier behind the counted wait; WAR: refill behind a barrier every reader waited in
    front of): the same bits every time, at a shape of several rounds of workgroups and at the ViT fc1 shape"""
    from valor_amd import kernels as Kn
    for M, N, K in ((8832, I, W), (100864, I, W), (16512, W, I)):
        A, B = _mk((M, K), 18, dev), _mk((N, K), 19, dev, 0.05)
        c0 = Kn.gemm(A, B)
        ref = _m32(wide, lambda: Kn.gemm(A, B))
        assert torch.equal(c0, ref)
        for _ in range(10):
            assert torch.equal(Kn.gemm(A, B), c0)
