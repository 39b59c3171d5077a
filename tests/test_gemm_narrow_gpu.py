"""The 256 x 128 two-workgroups-per-CU 8-phase GEMM (csrc/gemm8n.hip, kernel family 4) against fp32 torch.matmul on the device, forced
onto every eligible problem (policy key 8 = 1) so that each of its code paths runs whatever the measured policy says: all four operand
layouts it serves, both schedules, one- / two- / many-K-tile contractions, M / N tails, the bf16 tile epilogue (plain, bias, activation,
pre-activation or derivative copy, saved-derivative multiply, C +=), the general fp32 epilogue (fp32 out, odd N), split-K with fused row
sums -- and at the bench's decoder / AST / ViT shapes, per 256 x 128 tile (a mis-rastered or dropped tile is invisible in a Frobenius norm
over thousands of tiles). Reference semantics: nn.Linear and its autograd GEMMs, model/bert.py:233-235,403-417, model/transformer.py:109-142,
model/clip.py:176-192."""
import pytest
import torch

pytestmark = pytest.mark.gpu

W, I = 768, 3072
TOL, TILE_TOL = 2.5e-3, 4e-3          # bf16 output rounding alone is ~1.1e-3 Frobenius (tests/test_gemm_bench_shapes_gpu.py)


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


@pytest.fixture(params=[(0, 0), (1, 0), (0, 1), (1, 1)], ids=["plain", "pipelined", "plain-mfma32", "pipelined-mfma32"])
def narrow(request, dev):
    """every eligible problem on the family-4 kernel, under one of its two schedules; -mfma32: the NN layout's main loop on
    v_mfma_f32_32x32x16_bf16 (policy key 9; the other layouts keep the 16 x 16 x 32 loop, their cases then simply run twice)"""
    from valor_amd import lib
    so = lib.load()
    sched, m32 = request.param
    old_p, old_s, old_m = so.valor_gemm_set_policy(8, 1), so.valor_gemm_set_narrow_sched(sched), so.valor_gemm_set_policy(9, m32)
    so.mfma32 = bool(m32)
    yield so
    so.valor_gemm_set_policy(8, old_p)
    so.valor_gemm_set_narrow_sched(old_s)
    so.valor_gemm_set_policy(9, old_m)


def test_two_workgroups_per_cu(dev):
    """80 KiB of LDS and <= 256 VGPRs per workgroup: the runtime must admit exactly the two workgroups per CU the kernel is built around
    (with policy key 9 set the query covers the 32 x 32 x 16 instantiations too)"""
    from valor_amd import lib
    so = lib.load()
    assert so.valor_gemm_narrow_occupancy() == 2
    old = so.valor_gemm_set_policy(9, 1)
    try:
        assert so.valor_gemm_narrow_occupancy() == 2
    finally:
        so.valor_gemm_set_policy(9, old)


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (256, 128, 64 * 3), (700, 328, 256), (1030, 1000, 448), (8832, 3 * W, W), (16512, W, I),
                                   (8832, 30522, W)])
def test_forward_nn(narrow, dev, M, N, K):
    """x.W^T + bias: one tile, tails in M and N (N % 8 != 0 takes the general epilogue), the decoder / AST shapes, the padded vocabulary"""
    from valor_amd import kernels as Kn
    assert narrow.valor_gemm_kernel_for(0, 0, 0, M, N, K, 0) == 4
    A, B, bias = _mk((M, K), 1, dev), _mk((N, K), 2, dev, 0.05), _mk((N,), 3, dev)
    ref = A.float() @ B.float().t()
    whole, worst = _tile_errors(Kn.gemm(A, B), ref)
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    whole, worst = _tile_errors(Kn.gemm(A, B, bias=bias), ref + bias.float())
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    c32 = Kn.gemm(A, B, bias=bias, out_dtype=torch.float32)
    assert float((c32 - ref - bias.float()).norm() / ref.norm()) < 2e-5


def test_forward_matches_the_other_families_bit_for_bit_on_plain_problems(narrow, dev):
    """same MFMA, same K order, same fp32 -> bf16 rounding: a plain product must equal the 128 x 128 kernel's to the bit"""
    from valor_amd import kernels as Kn
    M, N, K = 2048, 1152, 768
    A, B = _mk((M, K), 4, dev), _mk((N, K), 5, dev, 0.05)
    C4 = Kn.gemm(A, B)
    old = narrow.valor_gemm_set_policy(8, 0)
    try:
        C1 = Kn.gemm(A, B)
    finally:
        narrow.valor_gemm_set_policy(8, old)
    if narrow.mfma32:
        # 32 x 32 x 16 adds the k-octets of a K-tile in another order (16 k per instruction instead of 32): the fp32 sums differ in
        # their last bits, the bf16 results by at most one ulp on a small fraction of the elements
        d = (C4.float() - C1.float()).abs()
        assert float((d > 0).float().mean()) < 0.05 and bool((d <= 2.0 ** -7 * C1.float().abs() + 1e-30).all())
    else:
        assert torch.equal(C4, C1)


def test_fused_activation_epilogues(narrow, dev):
    """fc1 as ops.MlpFn launches it (bias + QuickGELU / erf-GELU, second output = act'(u) or u) and the fc2 dgrad that multiplies by it"""
    from valor_amd import kernels as Kn, lib
    M, N, K = 8832, I, W
    assert narrow.valor_gemm_kernel_for(0, 0, 0, M, N, K, 0) == 4
    A, B, bias = _mk((M, K), 6, dev), _mk((N, K), 7, dev, 0.05), _mk((N,), 8, dev, 0.5)
    u = A.float() @ B.float().t() + bias.float()
    sg = torch.sigmoid(1.702 * u)
    h, d = Kn.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True)
    for got, want in ((h, u * sg), (d, sg * (1 + 1.702 * u * (1 - sg)))):
        whole, worst = _tile_errors(got, want)
        assert whole < TOL and worst < TILE_TOL, (whole, worst)
    # tails in M and N (and a last tile column that is half empty) with two outputs
    At, Bt, bt = _mk((700, K), 61, dev), _mk((328, K), 62, dev, 0.05), _mk((328,), 63, dev, 0.5)
    ut = At.float() @ Bt.float().t() + bt.float()
    st = torch.sigmoid(1.702 * ut)
    ht, dt = Kn.gemm(At, Bt, bias=bt, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True)
    for got, want in ((ht, ut * st), (dt, st * (1 + 1.702 * ut * (1 - st)))):
        whole, worst = _tile_errors(got, want)
        assert whole < TOL and worst < TILE_TOL, (whole, worst)
    h2, u2 = Kn.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU, want_preact=True)
    assert torch.equal(h2, h)
    whole, worst = _tile_errors(u2, u)
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    g = Kn.gemm(A, B, bias=bias, act=lib.ACT_GELU_ERF)
    whole, worst = _tile_errors(g, torch.nn.functional.gelu(u))
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    # dgrad dY.W2 * act'(u) with the saved derivative (light epilogue: bf16 tile pass) and with the saved pre-activation (general epilogue)
    dY, W2 = _mk((M, W), 9, dev), _mk((W, I), 10, dev, 0.05)
    prod = dY.float() @ W2.float()
    assert narrow.valor_gemm_kernel_for(0, 0, 1, M, I, W, 0) == 4
    dU = Kn.gemm(dY, W2, trans_b=True, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, dact_aux=d)
    whole, worst = _tile_errors(dU, prod * d.float())
    assert whole < 1.5 * TOL and worst < 1.5 * TILE_TOL, (whole, worst)
    dU = Kn.gemm(dY, W2, trans_b=True, act=lib.ACT_QUICK_GELU, dact_aux=u2)
    uu = u2.float()
    s2 = torch.sigmoid(1.702 * uu)
    whole, worst = _tile_errors(dU, prod * (s2 * (1 + 1.702 * uu * (1 - s2))))
    assert whole < 1.5 * TOL and worst < 1.5 * TILE_TOL, (whole, worst)


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1030, 776, 320), (8832, W, I), (16512, I, W)])
def test_dgrad_nt_and_accumulate(narrow, dev, M, N, K):
    """dY.W (k-slow weight through the transposing LDS reads), plain and accumulating into an existing gradient"""
    from valor_amd import kernels as Kn
    assert narrow.valor_gemm_kernel_for(0, 0, 1, M, N, K, 0) == 4
    dY, Wt = _mk((M, K), 11, dev), _mk((K, N), 12, dev, 0.05)
    ref = dY.float() @ Wt.float()
    whole, worst = _tile_errors(Kn.gemm(dY, Wt, trans_b=True), ref)
    assert whole < TOL and worst < TILE_TOL, (whole, worst)
    acc0 = _mk((M, N), 13, dev)
    out = acc0.clone()
    Kn.gemm(dY, Wt, trans_b=True, out=out, accumulate=True)
    whole, worst = _tile_errors(out, ref + acc0.float())
    assert whole < 1.5 * TOL and worst < 1.5 * TILE_TOL, (whole, worst)


@pytest.mark.parametrize("Mo,No,Kt", [(W, W, 8832), (I, W, 8832), (W, I, 16512), (3 * W, W, 100864), (520, 392, 4096)])
def test_wgrad_tt_splitk_with_fused_row_sums(narrow, dev, Mo, No, Kt):
    """dY^T.X over the token rows: both operands k-slow, split-K over one round of 512 workgroup slots, the bias gradient as row sums on
    the matrix pipe, accumulated into existing buffers"""
    from valor_amd import kernels as Kn
    assert narrow.valor_gemm_kernel_for(0, 1, 1, Mo, No, Kt, 0) == 4
    dY, X = _mk((Kt, Mo), 14, dev, 0.1), _mk((Kt, No), 15, dev)
    ref = dY.float().t() @ X.float()
    rs_ref = dY.float().sum(dim=0)
    whole, worst = _tile_errors(Kn.gemm(dY, X, trans_a=True, trans_b=True), ref)
    assert whole < 1.5 * TOL and worst < 1.5 * TILE_TOL, (whole, worst)
    g0, b0 = _mk((Mo, No), 16, dev), _mk((Mo,), 17, dev)
    gw, gb = g0.clone(), b0.clone()
    assert Kn.gemm_fuses_rowsum(dY, X, True, True)
    Kn.gemm(dY, X, trans_a=True, trans_b=True, out=gw, accumulate=True, rowsum_out=gb, rowsum_accumulate=True)
    whole, worst = _tile_errors(gw, ref + g0.float())
    assert whole < 1.5 * TOL and worst < 1.5 * TILE_TOL, (whole, worst)
    want = rs_ref + b0.float()
    assert float((gb.float() - want).norm() / want.norm()) < 3e-3
    d32 = Kn.gemm(dY, X, trans_a=True, trans_b=True, out_dtype=torch.float32, splitk=False)      # no workspace: one slice, fp32 out
    assert float((d32 - ref).norm() / ref.norm()) < 2e-5


def test_repeated_launches_are_deterministic(narrow, dev):
    """race screen: the counted-vmcnt ring (RAW one phase after the wait, WAR two phases after the read) must give the same bits every
    time, also with other work on the device"""
    from valor_amd import kernels as Kn
    A, B = _mk((8832, W), 18, dev), _mk((I, W), 19, dev, 0.05)
    dY, Wt = _mk((8832, I), 20, dev), _mk((I, W), 21, dev, 0.05)
    c0, d0 = Kn.gemm(A, B), Kn.gemm(dY, Wt, trans_b=True)
    for _ in range(20):
        assert torch.equal(Kn.gemm(A, B), c0)
        assert torch.equal(Kn.gemm(dY, Wt, trans_b=True), d0)


def test_per_call_policy_selects_the_kernel_without_global_hooks(dev):
    """lib.GemmPolicy (valor_gemm_policy): the family-4 kernel, its 32 x 32 x 16 main loop and its plain schedule chosen for ONE call
    each -- the same results as under the process-global hooks, and the hooks' values untouched afterwards."""
    from valor_amd import kernels as Kn, lib
    so = lib.load()
    before = [so.valor_gemm_set_policy(k, -1) for k in range(12)] + [so.valor_gemm_set_narrow_sched(-1), so.valor_gemm_set_variant(-1)]
    M, N, K = 8832, 3 * W, W
    A, B, bias = _mk((M, K), 41, dev), _mk((N, K), 42, dev, 0.05), _mk((N,), 43, dev)
    ref = A.float() @ B.float().t() + bias.float()
    outs = {}
    for name, pol in (("narrow", lib.GemmPolicy.make(narrow=1)), ("narrow-mfma32", lib.GemmPolicy.make(narrow=1, mfma32=1)),
                      ("narrow-plain", lib.GemmPolicy.make(narrow=1, sched_narrow=0)), ("wide", lib.GemmPolicy.make(narrow=0)),
                      ("small", lib.GemmPolicy.make(variant=1))):
        import ctypes
        fam = so.valor_gemm_kernel_for_tuned(ctypes.addressof(pol), 0, 0, 0, M, N, K, 0)
        assert fam == {"narrow": 4, "narrow-mfma32": 4, "narrow-plain": 4, "wide": 3, "small": 1}[name]
        outs[name] = Kn.gemm(A, B, bias=bias, policy=pol)
        whole, worst = _tile_errors(outs[name], ref)
        assert whole < TOL and worst < TILE_TOL, (name, whole, worst)
    assert torch.equal(outs["narrow"], outs["narrow-plain"]) and torch.equal(outs["narrow"], outs["small"])
    after = [so.valor_gemm_set_policy(k, -1) for k in range(12)] + [so.valor_gemm_set_narrow_sched(-1), so.valor_gemm_set_variant(-1)]
    assert before == after
