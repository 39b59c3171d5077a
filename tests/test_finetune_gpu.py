"""SURVEY 8f row 4 on the GPU: the 'ret%..' / 'cap%..' finetune tasks (config/fast-retrieval-*.json, caption-*.json) and caption
generation (greedy and beam-3) of the native model against
  (1) the CPU oracle on tiny models (every variant, task prompt on and off), and
  (2) golden vectors from the UNMODIFIED reference at base widths (tests/golden/ref_*_ft.pt, oracle/make_goldens.py run_finetune).
fp32: losses within 1e-4, generated token ids identical. bf16: losses within the north-star 1e-3 (5e-3 for the B = 2 contrastive loss,
DESIGN.md section 4), generated ids identical up to the first step whose reference margin is inside the bf16 logit error."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


def _build(spec, sd, dtype, dev, **opts):
    from valor_amd.model.valor import VALOR
    m = VALOR({"dropout": 0.0, "drop_path_rate": 0.0, **opts}, spec=spec, dtype=dtype, device=dev)
    m.load_state_dict(sd, strict=True)
    m.train()
    return m


@pytest.mark.parametrize("variant,prompt", [("clip", False), ("swin", False), ("clip_bert", True)])
def test_tiny_fp32_finetune_tasks_and_generation_match_oracle(dev, variant, prompt):
    from valor_amd import synth
    import valor_oracle as VO
    spec = {"clip": synth.tiny_spec, "swin": synth.tiny_swin_spec, "clip_bert": synth.tiny_clip_bert_spec}[variant]()
    sd = synth.make_state_dict(spec, seed=5, w_std=0.05)
    batch = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=6)
    orc = VO.Oracle(spec, VO.trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab), use_task_prompt=prompt)
    model = _build(spec, sd, torch.float32, dev, use_task_prompt=prompt, beam_size=3, max_generation_len=8)
    # --- retrieval finetune loss + gradients reach the arena
    random.seed(1)
    o = orc.forward(batch, "ret%tva%tv", compute_loss=True)
    random.seed(1)
    n = model(batch, task="ret%tva%tv", compute_loss=True)
    assert set(n) == {"contra_loss"}
    assert abs(float(o["contra_loss"]) - float(n["contra_loss"])) <= 1e-4 * abs(float(o["contra_loss"]))
    n["contra_loss"].backward()
    assert float(model.arena.grad.float().abs().max()) > 0
    model.zero_grad()
    with torch.no_grad():
        oe, ne = orc.forward(batch, "ret%tva%tv", compute_loss=False), model(batch, task="ret%tva%tv", compute_loss=False)
    for k in ("feat_t", "feat_v", "feat_a"):
        assert torch.allclose(oe[k], ne[k].float().cpu(), atol=2e-5), k
    # --- caption finetune loss
    random.seed(2)
    o = orc.forward(batch, "cap%tva%tv", compute_loss=True)
    random.seed(2)
    n = model(batch, task="cap%tva%tv", compute_loss=True)
    assert set(n) == {"caption_loss"}
    assert abs(float(o["caption_loss"]) - float(n["caption_loss"])) <= 1e-4 * abs(float(o["caption_loss"]))
    # --- generation: greedy and beam search, every group
    with torch.no_grad():
        og = orc.forward_cap(batch, "cap%tva%tv%ta", compute_loss=False, beam_size=1, max_generation_len=8)
        ob = orc.forward_cap(batch, "cap%tva%tv%ta", compute_loss=False, beam_size=3, max_generation_len=8)
    model.beam_size = 1
    ng = model(batch, task="cap%tva%tv%ta", compute_loss=False)
    model.beam_size = 3
    nb = model(batch, task="cap%tva%tv%ta", compute_loss=False)
    assert model.training                      # generate_cap restores the mode it found
    for k in ("generated_sequences_t_va", "generated_sequences_t_v", "generated_sequences_t_a"):
        assert torch.equal(og[k], ng[k].cpu()), (k, og[k], ng[k])
        assert torch.equal(ob[k], nb[k].cpu()), (k, ob[k], nb[k])
    assert float(ng["logprobs_t_va"].abs().max()) == 0.0          # pretrain.py:992,1013: only 'sample' mode fills them


def _agree_until_margin(ref_seq, got_seq, margin, tol):
    """rows must agree up to (excluding) the first step whose reference margin is below tol"""
    n_checked = 0
    for r in range(ref_seq.shape[0]):
        small = (margin[r] < tol).nonzero()
        upto = int(small[0]) if small.numel() else ref_seq.shape[1]
        upto = min(upto, margin.shape[1])
        assert torch.equal(ref_seq[r, :upto], got_seq[r, :upto]), (r, upto, ref_seq[r], got_seq[r])
        n_checked += upto
    return n_checked


@pytest.mark.parametrize("name", ["ref_base_b2f2a1_ft", "ref_swin_b2f2a1_ft"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_base_widths_match_reference_goldens(dev, name, dtype):
    from test_model_gpu import _recipe_tensors
    g = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    rc = g["recipe"]
    spec, sd, batch = _recipe_tensors(rc)
    fp32 = dtype == torch.float32
    model = _build(spec, sd, dtype, dev, beam_size=3, max_generation_len=rc["max_generation_len"])
    with torch.no_grad():
        random.seed(rc["masker_seed"])
        ret = float(model(batch, task="ret%tva%tv", compute_loss=True)["contra_loss"])
        random.seed(rc["masker_seed"])
        cap = float(model(batch, task="cap%tva%tv", compute_loss=True)["caption_loss"])
    assert abs(ret - g["ret_loss"]) <= (1e-4 if fp32 else 5e-3) * abs(g["ret_loss"]), (ret, g["ret_loss"])
    assert abs(cap - g["cap_loss"]) <= (1e-4 if fp32 else 1e-3) * abs(g["cap_loss"]), (cap, g["cap_loss"])
    tol = 1e-4 if fp32 else 0.05            # logit error of the run: fp32 ~1e-5; bf16 ~2e-2 (tests/test_model_gpu.py BF16_TIE_BAND)
    model.beam_size = 1
    gr = model(batch, task="cap%tva%tv%ta", compute_loss=False)
    model.beam_size = 3
    bm = model(batch, task="cap%tva%tv", compute_loss=False)
    steps = _agree_until_margin(g["greedy"]["generated_sequences_t_va"], gr["generated_sequences_t_va"].cpu(), g["greedy_margin_t_va"], tol)
    steps += _agree_until_margin(g["beam3"]["generated_sequences_t_va"], bm["generated_sequences_t_va"].cpu(), g["beam3_gap_t_va"], tol)
    assert steps > 0
    if fp32:
        for k in g["greedy"]:
            assert torch.equal(g["greedy"][k], gr[k].cpu()), k
        for k in g["beam3"]:
            assert torch.equal(g["beam3"][k], bm[k].cpu()), k
    # rows that END: [SEP] bias raised by the recipe's delta, greedy only (finished beams tie in the reference's beam search)
    sd2 = dict(sd)
    sd2["cls.decoder.bias"] = sd["cls.decoder.bias"].clone()
    sd2["cls.decoder.bias"][102] += rc["eos_bias_delta"]
    model.load_state_dict(sd2, strict=True)
    model.beam_size = 1
    ge = model(batch, task="cap%tva%tv%ta", compute_loss=False)
    ref = g["greedy_eos"]["generated_sequences_t_va"]
    assert (ref == 102).any() and not (ref == 102).all()
    _agree_until_margin(ref, ge["generated_sequences_t_va"].cpu(), g["greedy_eos_margin_t_va"], tol)
    if fp32:
        for k in g["greedy_eos"]:
            assert torch.equal(g["greedy_eos"][k], ge[k].cpu()), k
