"""contrastive_fused.hip against the unfused HIP path (valor_gemm -> S, valor_fine_reduce_fwd / _bwd) and against fp64 torch math:
the fused forward keeps the [A, B, T, V] token similarities of model/pretrain.py:191-211 in registers, the backward builds d(sims) for a
chunk of texts at a time -- same outputs (score, directional maxima, first-arg-max bytes, gradients), no [B*T, B*Nv] tensor."""
import pytest
import torch
import torch.nn.functional as F

from test_contrastive_xent_gpu import _ref_fine_loss, _rel

pytestmark = pytest.mark.gpu


def _inputs(NA, NB, T, Nv, D, dev, seed=3, mask_b=False):
    g = torch.Generator().manual_seed(seed)
    fa = F.normalize(torch.randn((NA, T, D), generator=g), dim=-1).to(torch.bfloat16).to(dev)
    fb = F.normalize(torch.randn((NB, Nv, D), generator=g), dim=-1).to(torch.bfloat16).to(dev)
    lens = torch.randint(1, T + 1, (NA,), generator=g)
    maskA = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
    maskB = torch.ones((NB, Nv))
    if mask_b:                                         # vta / atv groups: the [text | audio] side carries the text mask
        lb = torch.randint(1, Nv + 1, (NB,), generator=g)
        maskB = (torch.arange(Nv)[None, :] < lb[:, None]).float()
    maskB = maskB.to(dev)
    wA = torch.softmax(torch.randn((NA, T), generator=g).to(dev).masked_fill(maskA == 0, float("-inf")), -1).contiguous()
    wB = torch.softmax(torch.randn((NB, Nv), generator=g).to(dev).masked_fill(maskB == 0, float("-inf")), -1).contiguous()
    return fa, fb, maskA, maskB, wA, wB


# (NA, NB, T, Nv, D): token paddings 16 / 32 / 64 on both sides, tile tails in both directions, a rectangular evaluation shape
SHAPES = [(67, 67, 32, 10, 512), (5, 5, 8, 2, 128), (130, 130, 32, 8, 512), (9, 9, 40, 34, 64), (33, 33, 8, 34, 512), (20, 7, 17, 17, 128),
          (3, 300, 32, 10, 512)]


@pytest.mark.parametrize("NA,NB,T,Nv,D", SHAPES)
def test_fused_forward_equals_the_unfused_kernels(dev, NA, NB, T, Nv, D):
    from valor_amd import kernels as K, lib
    from valor_amd.kernels import _ptr as p, _stream as st
    fa, fb, maskA, maskB, wA, wB = _inputs(NA, NB, T, Nv, D, dev, mask_b=Nv > 16)
    f32 = dict(dtype=torch.float32, device=dev)
    mk = lambda: (torch.full((NA, NB), float("nan"), **f32), torch.full((NA, NB, T), float("nan"), **f32), torch.full((NA, NB, Nv), float("nan"), **f32),
                  torch.full((NA, NB, T), 77, dtype=torch.uint8, device=dev), torch.full((NA, NB, Nv), 77, dtype=torch.uint8, device=dev))
    s1, a1, b1, ia1, ib1 = mk()
    lib.call("valor_fine_fused_fwd", st(), p(fa), p(fb), p(maskA), p(maskB), p(wA), p(wB), p(s1), p(a1), p(b1), p(ia1), p(ib1), NA, NB, T, Nv, D)
    # scores only (evaluation)
    s3 = torch.full((NA, NB), float("nan"), **f32)
    lib.call("valor_fine_fused_fwd", st(), p(fa), p(fb), p(maskA), p(maskB), p(wA), p(wB), p(s3), 0, 0, 0, 0, NA, NB, T, Nv, D)
    # reference 1: fp64 torch
    sims = torch.einsum("atd,bvd->abtv", fa.double(), fb.double()) * maskA.double()[:, None, :, None] * maskB.double()[None, :, None, :]
    a_ref, ia_ref = sims.max(dim=-1)
    b_ref, ib_ref = sims.max(dim=-2)
    s_ref = (torch.einsum("abt,at->ab", a_ref, wA.double()) + torch.einsum("abv,bv->ab", b_ref, wB.double())) / 2
    assert torch.allclose(a1.double(), a_ref, atol=2e-6) and torch.allclose(b1.double(), b_ref, atol=2e-6)
    assert torch.allclose(s1.double(), s_ref, atol=2e-6) and torch.equal(s1, s3)
    # the argmax bytes pick an entry that attains the maximum (fp64 and fp32 may order near-ties differently: compare through the values)
    pick_a = torch.gather(sims, 3, ia1.long().unsqueeze(-1)).squeeze(-1)
    pick_b = torch.gather(sims, 2, ib1.long().unsqueeze(2)).squeeze(2)
    assert torch.allclose(pick_a, a_ref, atol=2e-6) and torch.allclose(pick_b, b_ref, atol=2e-6)
    assert float((ia1.long() == ia_ref).float().mean()) > 0.995 and float((ib1.long() == ib_ref).float().mean()) > 0.995
    # reference 2: the unfused kernels (square shapes only: valor_fine_reduce_fwd is B x B)
    if NA == NB:
        ldS = (NB * Nv + 7) // 8 * 8
        S = torch.zeros((NA * T, ldS), **f32)
        K.gemm(fa.view(NA * T, D), fb.view(NB * Nv, D), out=S[:, :NB * Nv], out_dtype=torch.float32)
        s2, a2, b2, ia2, ib2 = mk()
        lib.call("valor_fine_reduce_fwd", st(), p(S), ldS, p(maskA), p(maskB), p(wA), p(wB), p(s2), p(a2), p(b2), p(ia2), p(ib2), NA, T, Nv)
        assert torch.allclose(s1, s2, atol=1e-6) and torch.allclose(a1, a2, atol=1e-6) and torch.allclose(b1, b2, atol=1e-6)
        assert float((ia1 == ia2).float().mean()) > 0.999 and float((ib1 == ib2).float().mean()) > 0.999
        # backward tile: d(sims) of a chunk of texts == the same rows of the unfused dense d(sims)
        dscore = torch.randn((NA, NB), **f32)
        for dt in (torch.bfloat16, torch.float32):
            dS_full = torch.zeros((NA * T, ldS), dtype=dt, device=dev)
            dwA, dwB = torch.empty((NA, T), **f32), torch.empty((NB, Nv), **f32)
            lib.call("valor_fine_reduce_bwd", st(), K.dt_of(dS_full), p(dscore), p(maskA), p(maskB), p(wA), p(wB), p(a2), p(b2), p(ia2), p(ib2),
                     p(dS_full), ldS, p(dwA), p(dwB), NA, T, Nv)
            a0, na = NA // 3, NA - NA // 3 - 1 if NA > 3 else 1
            dS_c = torch.full((na * T, ldS), 5.0, dtype=dt, device=dev)
            lib.call("valor_fine_ds_chunk", st(), K.dt_of(dS_c), p(dscore), p(maskA), p(maskB), p(wA), p(wB), p(ia2), p(ib2), p(dS_c), ldS,
                     a0, na, NB, T, Nv)
            assert torch.equal(dS_c[:, :NB * Nv], dS_full[a0 * T:(a0 + na) * T, :NB * Nv])
            assert bool((dS_c[:, NB * Nv:] == 5.0).all())
        dwA2, dwB2 = torch.empty((NA, T), **f32), torch.empty((NB, Nv), **f32)
        lib.call("valor_fine_weight_grad", st(), p(dscore), p(a2), p(b2), p(dwA2), p(dwB2), NA, T, Nv)
        assert torch.equal(dwA2, dwA) and torch.equal(dwB2, dwB)


@pytest.mark.parametrize("B,T,Nv,chunk", [(512, 32, 10, 64), (100, 32, 8, 32), (70, 40, 34, 64)])
def test_fused_autograd_path_matches_fp64_and_never_builds_the_pair_tensor(dev, B, T, Nv, chunk):
    """ops.fine_contrastive on bf16 features (the fused path) at the 8-GPU global batch: loss and every gradient against fp64; the peak
    memory of forward + backward stays far below one fp32 [B*T, B*Nv] tensor (B = 512: 335 MB) -- the unfused path allocates that
    and a dense d(sims) on top; several backward chunks (fp32 accumulation of dfeatB across chunks)."""
    from valor_amd import lib, ops
    D = 512
    g = torch.Generator().manual_seed(11)
    fa = F.normalize(torch.randn((B, T, D), generator=g), dim=-1).to(torch.bfloat16).to(dev)
    fb = F.normalize(torch.randn((B, Nv, D), generator=g), dim=-1)
    fb = F.normalize(fb + 0.5 * fa[:, torch.arange(Nv) % T].float().cpu(), dim=-1).to(torch.bfloat16).to(dev)
    wa_raw, wb_raw = torch.randn((B, T), generator=g).to(dev), torch.randn((B, Nv), generator=g).to(dev)
    lens = torch.randint(5, T + 1, (B,), generator=g)
    maskA = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
    maskB = torch.ones((B, Nv), device=dev)
    k = torch.tensor(14.285, device=dev)
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    old_chunk = ops._FINE_CHUNK
    ops._FINE_CHUNK = chunk
    try:
        assert lib.load().valor_fine_set_fused(-1) == 1
        a1, b1, wa1, wb1, k1 = leaf(fa), leaf(fb), leaf(wa_raw), leaf(wb_raw), leaf(k)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = ops.fine_contrastive(a1, b1, wa1, wb1, maskA, maskB, k1)
        loss.backward()
        torch.cuda.synchronize()
        extra = torch.cuda.max_memory_allocated() - base
    finally:
        ops._FINE_CHUNK = old_chunk
    pair_tensor = B * T * B * Nv * 4
    # what legitimately scales with B^2: A2B + B2A (fp32) and the argmax bytes = 5 (T + Nv) B^2 bytes, plus the chunk tile and gradients
    assert extra < 5 * (T + Nv) * B * B + 2 * chunk * T * B * Nv * 2 + 64 * (1 << 20), (extra, pair_tensor)
    if B >= 256:
        assert extra < 0.6 * pair_tensor, (extra, pair_tensor)
    a2, b2, wa2, wb2, k2 = [leaf(t.double()) for t in (fa, fb, wa_raw, wb_raw, k)]
    ref, _ = _ref_fine_loss(a2, b2, wa2, wb2, maskA.double(), maskB.double(), k2)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-3 * abs(float(ref)), (float(loss), float(ref))
    for name, got, want in (("dfeatA", a1.grad, a2.grad), ("dfeatB", b1.grad, b2.grad), ("dwA", wa1.grad, wa2.grad),
                            ("dwB", wb1.grad, wb2.grad), ("dk", k1.grad, k2.grad)):
        assert _rel(got, want) < 2e-2, (name, _rel(got, want))
    # and the two HIP paths agree with each other far more tightly than either does with fp64 (same bf16 inputs, fp32 accumulation)
    so = lib.load()
    so.valor_fine_set_fused(0)
    try:
        a3, b3, wa3, wb3, k3 = leaf(fa), leaf(fb), leaf(wa_raw), leaf(wb_raw), leaf(k)
        loss3 = ops.fine_contrastive(a3, b3, wa3, wb3, maskA, maskB, k3)
        loss3.backward()
    finally:
        so.valor_fine_set_fused(1)
    assert abs(float(loss) - float(loss3)) <= 1e-5 * abs(float(loss3))
    assert _rel(a1.grad, a3.grad) < 5e-3 and _rel(b1.grad, b3.grad) < 5e-3 and _rel(wa1.grad, wa3.grad) < 1e-4
