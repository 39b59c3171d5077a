"""Kernel-level parity of contrastive.hip (fine-grained MGA similarity reduction + InfoNCE, model/pretrain.py:191-211,
model/modeling.py:418-433) and xent.hip (softmax cross-entropy on the vocabulary, model/pretrain.py:444) against fp64 torch math,
at the GLOBAL batch of the 8-GPU configuration (B = 512 pairs per side, 32 text tokens, Nv = 10 video + audio tokens) -- the
model-level tests only ever see B <= 4."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_fine_loss(fa, fb, wa_raw, wb_raw, maskA, maskB, k):
    """compute_fine_matrix_slice + contrastive_loss in fp64 (masks multiply, max over v / over t, softmax token weights)"""
    wA = torch.softmax(wa_raw.masked_fill(maskA == 0, float("-inf")), dim=-1)
    wB = torch.softmax(wb_raw.masked_fill(maskB == 0, float("-inf")), dim=-1)
    logits = torch.einsum("atd,bvd->abtv", fa, fb)
    logits = logits * maskA[:, None, :, None] * maskB[None, :, None, :]
    a2b = logits.max(dim=-1)[0]
    b2a = logits.max(dim=-2)[0]
    score = (torch.einsum("abt,at->ab", a2b, wA) + torch.einsum("abv,bv->ab", b2a, wB)) / 2.0
    s = score * k
    l1 = (-F.log_softmax(s, dim=1)).diag()
    l2 = (-F.log_softmax(s, dim=0)).diag()
    return torch.mean(torch.cat((l1, l2), dim=0)), score


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dtype,B,Nv", [(torch.float32, 512, 10), (torch.bfloat16, 512, 10), (torch.float32, 67, 8), (torch.float32, 5, 2)])
def test_fine_contrastive_at_global_batch(dev, dtype, B, Nv):
    from valor_amd import ops
    T, D = 32, 512
    g = torch.Generator().manual_seed(7)
    fa = F.normalize(torch.randn((B, T, D), generator=g), dim=-1).to(dtype).to(dev)
    fb = F.normalize(torch.randn((B, Nv, D), generator=g), dim=-1).to(dtype).to(dev)
    # correlated pairs so the diagonal is meaningful and the max / argmax are not degenerate
    fb = F.normalize(fb.float() + 0.5 * fa[:, :Nv].float(), dim=-1).to(dtype)
    wa_raw = torch.randn((B, T), generator=g).to(dev)
    wb_raw = torch.randn((B, Nv), generator=g).to(dev)
    lens = torch.randint(5, T + 1, (B,), generator=g)
    maskA = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
    maskB = torch.ones((B, Nv), device=dev)
    k = torch.tensor(14.285, device=dev)
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    a1, b1, wa1, wb1, k1 = leaf(fa), leaf(fb), leaf(wa_raw), leaf(wb_raw), leaf(k)
    loss = ops.fine_contrastive(a1, b1, wa1, wb1, maskA, maskB, k1)
    loss.backward()
    a2, b2, wa2, wb2, k2 = [leaf(t.double()) for t in (fa, fb, wa_raw, wb_raw, k)]
    ref, _ = _ref_fine_loss(a2, b2, wa2, wb2, maskA.double(), maskB.double(), k2)
    ref.backward()
    ltol, gtol = (2e-6, 2e-5) if dtype == torch.float32 else (2e-3, 2e-2)
    assert abs(float(loss) - float(ref)) <= ltol * abs(float(ref)), (float(loss), float(ref))
    for name, got, want in (("dfeatA", a1.grad, a2.grad), ("dfeatB", b1.grad, b2.grad), ("dwA", wa1.grad, wa2.grad),
                            ("dwB", wb1.grad, wb2.grad), ("dk", k1.grad, k2.grad)):
        assert _rel(got, want) < gtol, (name, _rel(got, want))
    # padded text positions carry no gradient to their weights
    assert float((wa1.grad * (1 - maskA)).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xent_against_fp64(dev, dtype):
    """valor_xent_fwd / _bwd on the padded-vocabulary layout the prediction head uses (V = 30522 in rows of 30528), labels incl. the
    first and last class, upstream gradient read from the device"""
    from valor_amd import lib
    from valor_amd.kernels import _ptr, _stream, dt_of
    n, V, Vpad = 1999, 30522, 30528
    g = torch.Generator().manual_seed(9)
    buf = torch.full((n, Vpad), float("nan"), dtype=dtype, device=dev)
    logits = (3.0 * torch.randn((n, V), generator=g)).to(dtype)
    buf[:, :V] = logits.to(dev)
    labels = torch.randint(0, V, (n,), generator=g)
    labels[0], labels[1] = 0, V - 1
    labels_d = labels.to(dev)
    loss_rows = torch.empty(n, dtype=torch.float32, device=dev)
    lse = torch.empty(n, dtype=torch.float32, device=dev)
    lib.call("valor_xent_fwd", _stream(), dt_of(buf), _ptr(buf), _ptr(labels_d), _ptr(loss_rows), _ptr(lse), n, V, Vpad)
    x = logits.double().requires_grad_(True)
    ref_rows = F.cross_entropy(x, labels, reduction="none")
    tol = 2e-6 if dtype == torch.float32 else 2e-5          # the statistics are fp32 in both modes; the INPUT is what is rounded
    assert _rel(loss_rows.cpu(), ref_rows) < tol
    assert _rel(lse.cpu(), torch.logsumexp(x, dim=-1)) < tol
    up = torch.tensor(0.37, device=dev)
    (0.37 * ref_rows.mean()).backward()
    lib.call("valor_xent_bwd", _stream(), dt_of(buf), _ptr(buf), _ptr(labels_d), _ptr(lse), _ptr(up), 1.0 / n, n, V, Vpad)
    assert _rel(buf[:, :V].cpu(), x.grad) < (2e-6 if dtype == torch.float32 else 4e-3)
    assert float(buf[:, V:].float().abs().max()) == 0.0     # the ld padding is zero-filled (it is a GEMM operand next)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_label_smoothing_cross_entropy(dev, dtype):
    """valor_xent_smooth_fwd / _bwd (LabelSmoothing, model/pretrain.py:46-61): row loss = KL(smoothed target || softmax) and its gradient
    against fp64, on the padded-vocabulary layout; smoothing 0 reproduces valor_xent_fwd / _bwd bit for bit."""
    from valor_amd import lib
    from valor_amd.kernels import _ptr, _stream, dt_of
    n, V, Vpad, eps = 37, 30522, 30528, 0.1
    g = torch.Generator().manual_seed(2)
    logits = (3.0 * torch.randn((n, Vpad), generator=g)).to(dtype)
    labels = torch.randint(0, V, (n,), generator=g)
    buf = logits.to(dev).clone()
    lab_d = labels.to(dev)
    loss_rows = torch.empty(n, device=dev); lse = torch.empty(n, device=dev)
    lib.call("valor_xent_smooth_fwd", _stream(), dt_of(buf), _ptr(buf), _ptr(lab_d), _ptr(loss_rows), _ptr(lse), n, V, Vpad, eps)
    z = logits[:, :V].double().requires_grad_(True)
    logp = torch.log_softmax(z, -1)
    tgt = torch.full_like(logp, eps / (V - 1)); tgt.scatter_(1, labels.unsqueeze(1), 1.0 - eps)
    ref_rows = (tgt * (tgt.log() - logp)).sum(1)
    assert torch.allclose(loss_rows.cpu().double(), ref_rows.detach(), rtol=2e-4, atol=2e-4), (loss_rows[:4], ref_rows[:4])
    ref_rows.mean().backward()
    up = torch.full((), 1.0, device=dev)
    lib.call("valor_xent_smooth_bwd", _stream(), dt_of(buf), _ptr(buf), _ptr(lab_d), _ptr(lse), _ptr(up), 1.0 / n, n, V, Vpad, eps)
    got = buf[:, :V].cpu().double()
    tol = 2e-6 if dtype == torch.float32 else 2e-2
    assert float((got - z.grad).norm() / z.grad.norm()) < tol
    assert float(buf[:, V:].float().abs().max()) == 0.0
    # smoothing 0 == the plain entry points, bit for bit
    a, b = logits.to(dev).clone(), logits.to(dev).clone()
    la, lb = torch.empty(n, device=dev), torch.empty(n, device=dev)
    sa, sb = torch.empty(n, device=dev), torch.empty(n, device=dev)
    lib.call("valor_xent_smooth_fwd", _stream(), dt_of(a), _ptr(a), _ptr(lab_d), _ptr(la), _ptr(sa), n, V, Vpad, 0.0)
    lib.call("valor_xent_fwd", _stream(), dt_of(b), _ptr(b), _ptr(lab_d), _ptr(lb), _ptr(sb), n, V, Vpad)
    assert torch.equal(la, lb) and torch.equal(sa, sb)
