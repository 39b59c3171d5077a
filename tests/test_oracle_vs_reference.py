"""Pins the restated CPU oracle (oracle/valor_oracle.py) against the UNMODIFIED reference imported from
/root/reference (only present in the build container; skipped elsewhere, where tests/golden/ takes over)."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_harness  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="/root/reference not present")

TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"


@pytest.fixture(scope="module")
def setup():
    from valor_amd import synth
    from valor_oracle import Oracle
    spec = synth.base_spec()
    sd = synth.make_state_dict(spec, seed=50)
    ref = ref_harness.build_reference(state_dict=sd, dropout=0.0)
    sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.decoder.weight"}
    sd_o["cls.decoder.weight"] = sd_o["multimodal_encoder.embeddings.word_embeddings.weight"]
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    batch = synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=32, seed=51)
    return spec, ref, orc, sd_o, batch


def test_losses_and_grads_match_reference(setup):
    spec, ref, orc, sd_o, batch = setup
    random.seed(50)
    ref_out = ref(batch, task=TASK, compute_loss=True)
    sum(ref_out.values()).backward()
    random.seed(50)
    orc_out = orc.forward_pt(batch, TASK, compute_loss=True)
    sum(orc_out.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        a, b = float(ref_out[k]), float(orc_out[k])
        assert abs(a - b) <= 2e-5 * abs(a), (k, a, b)
    ref_params = dict(ref.named_parameters())
    checked = 0
    for name, p in ref_params.items():
        if p.grad is None:
            assert sd_o[name].grad is None or float(sd_o[name].grad.abs().max()) == 0.0, name
            continue
        g = sd_o[name].grad
        assert g is not None, name
        # key biases have an analytically ZERO gradient (softmax shift invariance): compare those absolutely
        scale = max(float(p.grad.norm()), 1e-5 * p.grad.numel() ** 0.5)
        err = float((g - p.grad).norm()) / scale
        assert err < 2e-4, (name, err)
        checked += 1
    assert checked > 800


def test_eval_argmax_matches_reference(setup):
    spec, ref, orc, sd_o, batch = setup
    with torch.no_grad():
        random.seed(7)
        r = ref(batch, task=TASK, compute_loss=False)
        random.seed(7)
        o = orc.forward_pt(batch, TASK, compute_loss=False)
    for k in ("caption_scores_tva", "caption_scores_tv", "caption_scores_ta", "mlm_scores_tva"):
        assert torch.equal(r[k].argmax(-1), o[k].argmax(-1)), k
    assert torch.equal(r["txt_labels_caption"], o["txt_labels_caption"])
    assert torch.allclose(r["feat_t"], o["feat_t"], atol=1e-5)
