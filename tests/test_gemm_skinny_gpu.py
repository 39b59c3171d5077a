"""The few-row weight-streaming GEMM (csrc/gemm_skinny.hip, kernel family 5: the decoder GEMMs of caption generation with a K|V cache,
valor_amd/decode.py; nn.Linear on a handful of rows in the reference, model/bert.py:233-235,351,403-420) against fp64 torch.matmul on
the device: every covered contraction length, ragged M / N (rows and columns past the last whole 64 x 16 block), every epilogue it has
(bias, activation, alpha, fp32 output), strided operands; the policy (key 11: 0 for the process -- the training step keeps its kernels --,
384 per call from the inference paths, kernels.infer_policy()) and what falls back to the 128 x 128 kernels; and the same products on the
kernel it replaces (family 1; family 4 at M = 384) within the bf16 rounding of the output."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / max(b.norm().item(), 1e-6)).item()


def _act(x, act):
    if act == 1:
        return x * 0.5 * (1 + torch.erf(x / 2 ** 0.5))
    if act == 2:
        return x * torch.sigmoid(1.702 * x)
    if act == 3:
        return torch.relu(x)
    return x


CASES = [(128, 768, 768), (128, 2304, 768), (128, 3072, 768), (128, 768, 3072), (64, 30522, 768), (384, 768, 768), (2, 768, 768),
         (65, 40, 512), (100, 1000, 1024), (33, 130, 4096), (1, 16, 768),
         (384, 768, 3072), (250, 3072, 768), (193, 100, 512), (300, 520, 4096), (192, 30522, 768)]      # above 192 rows: 128-row blocks per workgroup


@pytest.mark.parametrize("M,N,Kd", CASES)
def test_skinny_matches_fp64(dev, M, N, Kd):
    from valor_amd import kernels as K, lib
    so, pol = lib.load(), K.infer_policy()
    assert so.valor_gemm_kernel_for_tuned(ctypes.addressof(pol), 0, 0, 0, M, N, Kd, 0) == 5
    assert so.valor_gemm_kernel_for(0, 0, 0, M, N, Kd, 0) in (1, 2, 3, 4)          # the process default: the training step's kernels
    A, B, bias = _mk((M, Kd), dev, 1), _mk((N, Kd), dev, 2, 0.05), _mk((N,), dev, 3)
    ref = A.double() @ B.double().t()
    C = K.gemm(A, B, policy=pol)
    assert C.dtype == torch.bfloat16 and _rel(C, ref) < 4e-3, _rel(C, ref)
    for act in (0, 1, 2, 3):
        C = K.gemm(A, B, bias=bias, act=act, alpha=0.5, out_dtype=torch.float32, policy=pol)
        want = _act(0.5 * ref + bias.double(), act)
        assert C.dtype == torch.float32 and _rel(C, want) < 2e-5, (act, _rel(C, want))      # fp32 accumulation, fp32 output: no bf16 rounding
    # the kernel it replaces, on the same operands: equal up to the order of the fp32 partial sums
    C1 = K.gemm(A, B, bias=bias, act=1, out_dtype=torch.float32)
    C5 = K.gemm(A, B, bias=bias, act=1, out_dtype=torch.float32, policy=pol)
    assert _rel(C5, C1) < 2e-5


def test_skinny_strided_operands_and_untouched_padding(dev):
    """A as a column block of a wider matrix (the q | k | v layout), C as a column block of a wider output: rows / columns outside stay untouched"""
    from valor_amd import kernels as K
    M, N, Kd = 70, 200, 768
    Abig, B = _mk((M, 3 * Kd), dev, 5), _mk((N, Kd), dev, 6, 0.05)
    A = Abig[:, Kd:2 * Kd]
    out = torch.full((M, N + 56), 7.0, dtype=torch.bfloat16, device=dev)
    K.gemm(A, B, out=out[:, 8:8 + N], policy=K.infer_policy())
    ref = A.double() @ B.double().t()
    assert _rel(out[:, 8:8 + N], ref) < 4e-3
    assert bool((out[:, :8] == 7).all()) and bool((out[:, 8 + N:] == 7).all())


def test_skinny_policy_and_fallbacks(dev):
    from valor_amd import kernels as K, lib
    so, pol = lib.load(), K.infer_policy()
    fam = lambda M, N, Kd, ta=0, tb=0: so.valor_gemm_kernel_for_tuned(ctypes.addressof(pol), 0, ta, tb, M, N, Kd, 0)
    assert fam(384, 768, 768) == 5 and fam(385, 768, 768) != 5            # key 11: M <= 384
    assert fam(128, 768, 832) != 5 and fam(128, 768, 2048) != 5            # contraction lengths it has no instantiation for
    assert fam(128, 768, 768, 1, 1) != 5 and fam(128, 768, 768, 0, 1) != 5 # NN layout only
    assert so.valor_gemm_kernel_for_tuned(ctypes.addressof(pol), 1, 0, 0, 128, 768, 768, 0) == 0        # fp32: the register-staged kernel
    old = so.valor_gemm_set_policy(11, 200)                                # a process default (env VALOR_GEMM_SKINNY_ALL)
    try:
        assert so.valor_gemm_kernel_for(0, 0, 0, 128, 768, 768, 0) == 5 and so.valor_gemm_kernel_for(0, 0, 0, 256, 768, 768, 0) != 5
    finally:
        so.valor_gemm_set_policy(11, old)
    # epilogues it does not have run on the 128 x 128 kernels and give the same numbers: C +=, a pre-activation copy
    A, B = _mk((128, 768), dev, 7), _mk((768, 768), dev, 8, 0.05)
    ref = A.double() @ B.double().t()
    C = torch.ones((128, 768), dtype=torch.float32, device=dev)
    K.gemm(A, B, out=C, accumulate=True, out_dtype=torch.float32, policy=pol)
    assert _rel(C, ref + 1) < 2e-5


@pytest.mark.parametrize("M,N,Kd", [(128, 3072, 768), (384, 2304, 768), (384, 778, 3072), (250, 30522, 768), (65, 40, 512), (193, 100, 4096), (37, 530, 1024)])
def test_skinny_tiles_are_bit_identical(dev, monkeypatch, M, N, Kd):
    """Every workgroup tile of the family (64 / 128 rows x 16 / 32 / 64 columns; VALOR_SKINNY_TILE pins one per call, the launcher picks by
    grid size otherwise) gives every accumulator the same products in the same order and adds the eight partial tiles in wave order:
    identical bits, ragged rows / columns and strided outputs included; the default choice is one of them."""
    from valor_amd import kernels as K
    pol = K.infer_policy()
    A, B, bias = _mk((M, Kd), dev, 11), _mk((N, Kd), dev, 12, 0.05), _mk((N,), dev, 13)
    ref = _act(A.double() @ B.double().t() + bias.double(), 1)
    outs = {}
    for tile in ("4,1", "8,1", "4,2", "8,2", "4,4", None):
        if tile is None:
            monkeypatch.delenv("VALOR_SKINNY_TILE", raising=False)
        else:
            monkeypatch.setenv("VALOR_SKINNY_TILE", tile)
        wide = torch.full((M, N + 24), 7.0, dtype=torch.float32, device=dev)
        K.gemm(A, B, bias=bias, act=1, out=wide[:, 8:8 + N], out_dtype=torch.float32, policy=pol)
        assert bool((wide[:, :8] == 7).all()) and bool((wide[:, 8 + N:] == 7).all()), tile
        outs[tile] = (wide[:, 8:8 + N].clone(), K.gemm(A, B, policy=pol))
    assert _rel(outs["4,1"][0], ref) < 2e-5
    for tile, (c32, c16) in outs.items():
        assert torch.equal(c32, outs["4,1"][0]) and torch.equal(c16, outs["4,1"][1]), tile
