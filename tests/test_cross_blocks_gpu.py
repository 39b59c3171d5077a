"""cross_attn_type in {'va_parallel', 'video_audio', 'audio_video'} (model/bert.py:430-436,459-496: a cross-attention block per modality --
cross_attn_v / cross_attn_a with their own query / key / value / output / LayerNorm -- summed or applied one after the other; a group with
one modality runs that modality's block alone) on the native model against the CPU oracle, which tests/test_oracle_vs_reference.py pins on
the unmodified reference for the same three modes. fp32: pretraining losses 1e-4 and every gradient (both blocks' parameters included),
evaluation argmax ids, the caption finetune loss, greedy and beam-3 captions for every group, video QA answers; the state dict keeps the
reference's keys; bf16 losses at the north-star's 1e-3 at base widths on identical tensors."""
import dataclasses
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva%ta"
MODES = ["va_parallel", "video_audio", "audio_video"]


def _native(spec, sd, dtype, dev, **opts):
    from valor_amd.model.valor import VALOR
    m = VALOR({"dropout": 0.0, "drop_path_rate": 0.0, **opts}, spec=spec, dtype=dtype, device=dev)
    m.load_state_dict(sd, strict=True)
    m.train()
    return m


@pytest.mark.parametrize("mode", MODES)
def test_tiny_fp32_blocks_match_oracle(dev, mode):
    from test_model_gpu import _native_grads
    from valor_amd import synth
    import valor_oracle as VO
    spec = dataclasses.replace(synth.tiny_spec(), cross_attn_type=mode)
    sd = synth.make_state_dict(spec, seed=3, w_std=0.05)
    batch = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=4)
    sd_o = VO.trainable_copy(sd)
    orc = VO.Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    model = _native(spec, sd, torch.float32, dev, beam_size=3, max_generation_len=6)
    assert model.spec.cross_attn_type == mode
    assert set(model.state_dict().keys()) == set(sd.keys())                     # the reference's keys, cross_attn_v / cross_attn_a included
    random.seed(11); o_out = orc.forward_pt(batch, TASK, compute_loss=True); sum(o_out.values()).backward()
    random.seed(11); n_out = model(batch, task=TASK, compute_loss=True); sum(n_out.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        a, b = float(o_out[k].detach()), float(n_out[k].detach())
        assert abs(a - b) <= 1e-4 * abs(a), (k, a, b)
    ng = _native_grads(model)
    bad, blocks = [], 0
    for k, p in sd_o.items():
        if VO.is_alias_key(k) or not p.is_floating_point():
            continue
        go, gn = p.grad, ng[k].detach().cpu()
        if go is None:
            assert float(gn.abs().max()) == 0.0, k
            continue
        scale = max(float(go.norm()), 1e-5 * go.numel() ** 0.5)
        err = float((gn.reshape(go.shape) - go).norm()) / scale
        if err > 2e-3:
            bad.append((k, err))
        blocks += ("cross_attn_v." in k) or ("cross_attn_a." in k)
    assert not bad, bad[:10]
    assert blocks == 2 * spec.layers * 10
    model.zero_grad()
    with torch.no_grad():
        random.seed(12); oe = orc.forward_pt(batch, TASK, compute_loss=False)
        random.seed(12); ne = model(batch, task=TASK, compute_loss=False)
        for k in oe:
            if "scores" in k:
                assert torch.equal(oe[k].argmax(-1), ne[k].argmax(-1).cpu()), k
        random.seed(13); oc = orc.forward(batch, "cap%tva%tv%ta", compute_loss=True)
        random.seed(13); nc = model(batch, task="cap%tva%tv%ta", compute_loss=True)
        assert abs(float(oc["caption_loss"]) - float(nc["caption_loss"])) <= 1e-4 * abs(float(oc["caption_loss"]))
        og = orc.forward_cap(batch, "cap%tva%tv%ta", compute_loss=False, beam_size=1, max_generation_len=6)
        ob = orc.forward_cap(batch, "cap%tva%tv%ta", compute_loss=False, beam_size=3, max_generation_len=6)
    model.beam_size = 1
    ngr = model(batch, task="cap%tva%tv%ta", compute_loss=False)
    model.beam_size = 3
    nbm = model(batch, task="cap%tva%tv%ta", compute_loss=False)
    for k in ("generated_sequences_t_va", "generated_sequences_t_v", "generated_sequences_t_a"):
        assert torch.equal(og[k], ngr[k].cpu()), (k, og[k], ngr[k])
        assert torch.equal(ob[k], nbm[k].cpu()), (k, ob[k], nbm[k])
    qb = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=10, seed=7, questions=True)
    random.seed(3); oq = orc.forward(qb, "qa%tva%tv", compute_loss=True)
    random.seed(3); nq = model(qb, task="qa%tva%tv", compute_loss=True)
    assert abs(float(oq["qa_loss"]) - float(nq["qa_loss"])) <= 1e-4 * abs(float(oq["qa_loss"]))
    nq["qa_loss"].backward()
    model.zero_grad()
    with torch.no_grad():
        oa = orc.forward_qa(qb, "qa%tva%ta", compute_loss=False, beam_size_qa=1, max_generation_len=5)
    model.max_generation_len, model.beam_size_qa = 5, 1
    na = model(qb, task="qa%tva%ta", compute_loss=False)
    for k in ("generated_answers_t_va", "generated_answers_t_a"):
        assert torch.equal(oa[k], na[k].cpu()), k


@pytest.mark.parametrize("mode", ["va_parallel", "audio_video"])
def test_base_widths_bf16_blocks_meet_the_north_star(dev, mode):
    """base widths, 2-layer stacks, bf16-representable weights and inputs on both sides, batch 8: the three pretraining losses within 1e-3"""
    from valor_amd import synth
    import valor_oracle as VO
    spec = dataclasses.replace(synth.shallow_base_spec("clip"), cross_attn_type=mode)
    sd = synth.make_state_dict(spec, seed=5, w_std=0.02, bf16_exact=True)
    batch = synth.make_batch(spec, batch=8, frames=2, audio_slices=1, txt_len=32, seed=6, bf16_exact=True)
    orc = VO.Oracle(spec, VO.trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab))
    model = _native(spec, sd, torch.bfloat16, dev)
    with torch.no_grad():
        random.seed(1); o = orc.forward_pt(batch, TASK, compute_loss=True)
    random.seed(1); n = model(batch, task=TASK, compute_loss=True)
    sum(n.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        a, b = float(o[k]), float(n[k].detach())
        assert abs(a - b) <= 1e-3 * abs(a), (k, a, b)
    assert float(model.arena.grad.float().abs().max()) > 0


def test_unknown_cross_attn_type_is_refused(dev):
    from valor_amd import synth
    from valor_amd.model.valor import VALOR
    with pytest.raises(NotImplementedError):
        VALOR({"cross_attn_type": "audio_first"}, spec=synth.tiny_spec(), dtype=torch.float32, device=dev)


def test_training_steps_with_blocks_under_the_engine(dev):
    """five optimisation steps (dropout on, graphed encoders -- the decoder of these modes runs per pass, eagerly) of a 'va_parallel' model:
    finite losses, both blocks' weights move, the loss of the fixed batch falls"""
    from types import SimpleNamespace
    from valor_amd import ops, synth
    from valor_amd.engine import TrainEngine
    from valor_amd.model.valor import VALOR
    spec = dataclasses.replace(synth.tiny_spec(), cross_attn_type="va_parallel")
    sd = synth.make_state_dict(spec, seed=3, w_std=0.05)
    model = VALOR({"dropout": 0.1}, spec=spec, dtype=torch.bfloat16, device=dev)
    model.load_state_dict(sd, strict=True)
    opts = SimpleNamespace(learning_rate=2e-3, weight_decay=0.01, clip_lr=1e-4, clip_lr_text=1e-4, new_lr=0.0, decoder_lr=-1, betas=[0.9, 0.98],
                           warmup_ratio=0.0, num_train_steps=100, scheduler="warmup_linear", grad_norm=5.0, alloc_headroom_mb=0)
    eng = TrainEngine(model, opts, manage_gc=False, graphs=True)
    eng.optimizer.init_master_from(sd)
    batch = synth.make_batch(spec, batch=4, frames=2, audio_slices=1, txt_len=32, seed=4)
    w0 = {k: model.P[f"multimodal_encoder.encoder.layer.1.cross_attn_{k}.cross.kv.weight"].detach().float().clone() for k in "va"}
    try:
        random.seed(5)
        losses = []
        for _ in range(5):
            out = eng.train_step(batch, TASK)
            losses.append(sum(float(v) for v in out.values()))
        torch.cuda.synchronize()
        assert all(l == l and abs(l) < 1e4 for l in losses), losses
        assert losses[-1] < losses[0], losses
        for k in "va":
            w = model.P[f"multimodal_encoder.encoder.layer.1.cross_attn_{k}.cross.kv.weight"].detach().float()
            assert float((w - w0[k]).abs().max()) > 0, k
        assert "decoder" not in model._graph_segs and {"vit", "ast", "clip_text"} <= set(model._graph_segs)
    finally:
        model.enable_graphs(False)
        eng.close()
        ops.DropoutState.disable_device_base()
        ops.DropoutState.reset(1234)
