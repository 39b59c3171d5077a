"""Embedding sum and token assembly (misc.hip: valor_embed_*, valor_assemble_tokens_*, valor_sum_over_batch) against torch autograd of
the reference expressions -- words + positions + type vector (model/bert.py:211-215, model/clip.py:377-379) and [cls ; patches + bias] + pos
(model/clip.py:264-265, model/modeling.py:755-760) -- on both gradient routes: returned to autograd (plain leaves), and accumulated
straight into an arena gradient slot that already holds something (ops.GradSink: what the model does; repeated ids, a position table
longer than the sequence, the pad id occurring thousands of times)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _leaf(t, sink, name, g):
    """a parameter-like leaf; with `sink` it looks like an arena parameter whose gradient slot already holds `g0`"""
    p = t.clone().requires_grad_(True)
    g0 = None
    if sink:
        g0 = (0.5 * torch.randn(t.shape, generator=g)).to(t.dtype).to(t.device)
        p._arena_name = name
        p.grad = g0.clone()
    return p, g0


@pytest.mark.parametrize("sink", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,V,E,maxpos", [(64, 33, 1000, 768, 512), (3, 7, 50, 64, 7), (130, 40, 30522, 768, 512)])
def test_embedding_sum(dev, dtype, sink, B, L, V, E, maxpos):
    from valor_amd import ops
    g = torch.Generator().manual_seed(B * L)
    ids = torch.randint(1, V, (B, L), generator=g)
    ids[:, L // 2:] = 0                                   # the pad id: half of all positions
    ids[0, :3] = ids[1, :3]                               # repeated ids across rows
    ids = ids.to(dev)
    word0 = torch.randn((V, E), generator=g).to(dtype).to(dev)
    pos0 = torch.randn((maxpos, E), generator=g).to(dtype).to(dev)
    typ0 = torch.randn((E,), generator=g).to(dtype).to(dev)
    dout = torch.randn((B, L, E), generator=g).to(dtype).to(dev)

    word, gw = _leaf(word0, sink, "w", g)
    pos, gp = _leaf(pos0, sink, "p", g)
    typ, gt = _leaf(typ0, sink, "t", g)
    old = ops.GradSink.listener
    seen = []
    ops.GradSink.listener = seen.append
    try:
        out = ops.embed(ids, word, pos, typ, L)
        out.backward(dout)
    finally:
        ops.GradSink.listener = old
    torch.cuda.synchronize()

    wr, pr, tr = (t.double().requires_grad_(True) for t in (word0, pos0, typ0))
    ref = wr[ids] + pr[:L][None] + tr
    ref.backward(dout.double())
    tol = 1e-5 if dtype == torch.float32 else 1.2e-2
    assert _rel(out, ref) < (1e-6 if dtype == torch.float32 else 4e-3)
    for p_, g0, r, n in ((word, gw, wr, "w"), (pos, gp, pr, "p"), (typ, gt, tr, "t")):
        want = r.grad + (g0.double() if sink else 0.0)
        assert _rel(p_.grad, want) < tol, (n, _rel(p_.grad, want))
    if sink:
        assert sorted(seen) == ["p", "t", "w"]
        assert bool((word.grad[ids.unique()[-1] + 1:] == gw[ids.unique()[-1] + 1:]).all())      # rows of ids that do not occur: untouched


@pytest.mark.parametrize("sink", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Pn,E,bias", [(512, 196, 768, False), (128, 128, 768, True), (5, 3, 64, True), (9, 196, 1024, False)])
def test_token_assembly(dev, dtype, sink, N, Pn, E, bias):
    from valor_amd import ops
    g = torch.Generator().manual_seed(N + Pn)
    patches0 = torch.randn((N * Pn, E), generator=g).to(dtype).to(dev)
    cls0 = torch.randn((E,), generator=g).to(dtype).to(dev)
    pos0 = torch.randn((Pn + 1, E), generator=g).to(dtype).to(dev)
    b0 = torch.randn((E,), generator=g).to(dtype).to(dev) if bias else None
    dout = torch.randn((N, Pn + 1, E), generator=g).to(dtype).to(dev)

    patches = patches0.clone().requires_grad_(True)
    cls, gc = _leaf(cls0, sink, "c", g)
    pos, gp = _leaf(pos0, sink, "p", g)
    bb, gb = _leaf(b0, sink, "b", g) if bias else (None, None)
    out = ops.assemble_tokens(patches, cls, pos, bb, N, Pn)
    out.backward(dout)
    torch.cuda.synchronize()

    pr, cr, qr = (t.double().requires_grad_(True) for t in (patches0, cls0, pos0))
    br = b0.double().requires_grad_(True) if bias else None
    body = pr.view(N, Pn, E) + (br if bias else 0.0)
    ref = torch.cat((cr.expand(N, 1, E), body), dim=1) + qr
    ref.backward(dout.double())
    assert _rel(out, ref) < (1e-6 if dtype == torch.float32 else 4e-3)
    tol = 1e-5 if dtype == torch.float32 else 1.2e-2
    assert _rel(patches.grad, pr.grad) < tol
    for p_, g0, r, n in ((cls, gc, cr, "c"), (pos, gp, qr, "p")) + (((bb, gb, br, "b"),) if bias else ()):
        want = r.grad + (g0.double() if sink else 0.0)
        assert _rel(p_.grad, want) < tol, (n, _rel(p_.grad, want))
