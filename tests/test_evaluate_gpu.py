"""Evaluation slice (valor_amd/evaluate.py): the rectangular / sliced fine-grained score matrix (pretrain.py:178-211) and validate_pt's
masked-token accuracies and retrieval recall (test.py:404-665, 714-774) against the CPU oracle on the same weights and batches."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu
TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"


def test_rectangular_sliced_fine_matrix(dev):
    import valor_oracle as VO
    from valor_amd.evaluate import compute_fine_matrix
    g = torch.Generator().manual_seed(3)
    NA, NB, T, Nv, D = 37, 1301, 32, 10, 128                 # NB > 1200: the reference's slicing branch (slices of 100 rows of A)
    fa = torch.nn.functional.normalize(torch.randn((NA, T, D), generator=g), dim=-1)
    fb = torch.nn.functional.normalize(torch.randn((NB, Nv, D), generator=g), dim=-1)
    lens = torch.randint(4, T + 1, (NA,), generator=g)
    maskA = (torch.arange(T)[None] < lens[:, None]).long()
    maskB = torch.ones((NB, Nv), dtype=torch.long)
    wA, wB = torch.randn((NA, T), generator=g), torch.randn((NB, Nv), generator=g)
    want = VO.Oracle.compute_fine_matrix(fa, fb, maskA, maskB, wA, wB)
    for kw in ({}, dict(slice_rows=8, slice_above=0)):
        got = compute_fine_matrix(fa.to(dev), fb.to(dev), maskA.to(dev), maskB.to(dev), wA.to(dev), wB.to(dev), **kw)
        assert got.shape == (NA, NB)
        assert torch.allclose(got.cpu(), want, atol=2e-5, rtol=1e-5), float((got.cpu() - want).abs().max())


def test_validate_pt_matches_oracle(dev):
    import valor_oracle as VO
    from valor_amd import synth
    from valor_amd.evaluate import compute_metric_ret, validate_pt
    from valor_amd.model.valor import VALOR
    spec = synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=3, w_std=0.05)
    batches = []
    for i in range(3):
        b = synth.make_batch(spec, batch=4, frames=2, audio_slices=1, txt_len=32, seed=10 + i)
        b["ids"] = [f"v{4 * i + j}" for j in range(4)]
        b["ids_txt"] = list(b["ids"])
        batches.append(b)
    model = VALOR({"dropout": 0.0}, spec=spec, dtype=torch.float32, device=dev)
    model.load_state_dict(sd, strict=True)
    random.seed(21)
    log = validate_pt(model, batches, TASK)
    # the same through the oracle
    orc = VO.Oracle(spec, sd, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    random.seed(21)
    hits, words = {}, {"caption": 0, "mlm": 0}
    feats = {"feat_t": [], "feat_v": [], "feat_a": [], "txt_tokens": []}
    with torch.no_grad():
        for b in batches:
            ev = orc.forward_pt(b, TASK, compute_loss=False)
            for k in feats:
                feats[k].append(ev[k])
            for tag in ("caption", "mlm"):
                lab = ev[f"txt_labels_{tag}"]; lab = lab[lab != -1]
                words[tag] += lab.numel()
                for g in ("tva", "tv", "ta"):
                    if f"{tag}_scores_{g}" in ev:
                        hits[f"{tag}_{g}"] = hits.get(f"{tag}_{g}", 0) + int((ev[f"{tag}_scores_{g}"].argmax(-1) == lab).sum())
        ft, fv, fa, tok = (torch.cat(feats[k], 0) for k in ("feat_t", "feat_v", "feat_a", "txt_tokens"))
        maskA = (tok != 0).long()
        ids = [x for b in batches for x in b["ids"]]
        fw = orc.fine_weight
        want = {"t2v_recall": compute_metric_ret(orc.compute_fine_matrix(ft, fv, maskA, torch.ones(*fv.shape[:2]).long(), fw("text", ft), fw("video", fv)), ids, ids)["forward_recall"],
                "t2a_recall": compute_metric_ret(orc.compute_fine_matrix(ft, fa, maskA, torch.ones(*fa.shape[:2]).long(), fw("text", ft), fw("audio", fa)), ids, ids)["forward_recall"]}
        fva = torch.cat((fv, fa), 1)
        want["t2va_recall"] = compute_metric_ret(orc.compute_fine_matrix(ft, fva, maskA, torch.ones(*fva.shape[:2]).long(), fw("text", ft),
                                                                         torch.cat((fw("video", fv), fw("audio", fa)), 1)), ids, ids)["forward_recall"]
    for tag in ("caption", "mlm"):
        for g in ("tva", "tv", "ta"):
            if f"{tag}_{g}" in hits:
                assert log[f"{tag}_acc_{g}"] == round(hits[f"{tag}_{g}"] / words[tag], 2), (tag, g)
    assert {"caption_acc_tva", "caption_acc_tv", "caption_acc_ta", "mlm_acc_tva"} <= set(log)
    for k, v in want.items():
        assert log[k] == v, (k, log[k], v)


@pytest.mark.parametrize("late", [False, True])
def test_validate_pt_with_coarse_features(dev, late):
    """contra_type='coarse' at evaluation (test.py:640-660): recall from the plain similarity matrices of the pooled features (tva through
    va_fusion, or the late-fusion sum) -- the recall strings of the oracle's features scored the same way."""
    import dataclasses
    import valor_oracle as VO
    from valor_amd import synth
    from valor_amd.evaluate import compute_metric_ret, validate_pt
    from valor_amd.model.valor import VALOR
    spec = dataclasses.replace(synth.tiny_spec(), contra_type="coarse", late_fusion=late)
    sd = synth.make_state_dict(spec, seed=3, w_std=0.05)
    batches = []
    for i in range(3):
        b = synth.make_batch(spec, batch=4, frames=2, audio_slices=1, txt_len=32, seed=20 + i)
        b["ids"] = [f"v{4 * i + j}" for j in range(4)]
        b["ids_txt"] = list(b["ids"])
        batches.append(b)
    model = VALOR({"dropout": 0.0}, spec=spec, dtype=torch.float32, device=dev)
    model.load_state_dict(sd, strict=True)
    task = "pt_contra%tva%tv%ta"
    log = validate_pt(model, batches, task)
    orc = VO.Oracle(spec, sd, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    feats = {"feat_t": [], "feat_v": [], "feat_a": []}
    with torch.no_grad():
        for b in batches:
            ev = orc.forward_pt(b, task, compute_loss=False)
            for k in feats:
                feats[k].append(ev[k])
        ft, fv, fa = (torch.cat(feats[k], 0) for k in ("feat_t", "feat_v", "feat_a"))
        assert ft.dim() == 2 and fv.dim() == 2
        ids = [x for b in batches for x in b["ids"]]
        if late:
            tva = ft @ fv.t() + ft @ fa.t()
        else:
            fva = torch.nn.functional.normalize(torch.nn.functional.linear(torch.cat((fv, fa), -1), sd["va_fusion.weight"], sd["va_fusion.bias"]), dim=-1)
            tva = ft @ fva.t()
        want = {"t2v_recall": compute_metric_ret(ft @ fv.t(), ids, ids)["forward_recall"], "t2a_recall": compute_metric_ret(ft @ fa.t(), ids, ids)["forward_recall"],
                "t2va_recall": compute_metric_ret(tva, ids, ids)["forward_recall"]}
    assert log == want, (log, want)
