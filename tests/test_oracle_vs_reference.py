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


@pytest.fixture(scope="module", params=["clip", "swin"])
def setup(request):
    """clip: config/pretrain-VALOR-base.json; swin: scripts/pretrain.sh:3-8 (VideoSwin-B + BERT text)"""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    if request.param == "swin":
        spec = synth.swin_spec()
        ropts = ref_harness.default_opts(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased")
    else:
        spec, ropts = synth.base_spec(), None
    sd = synth.make_state_dict(spec, seed=50)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0)
    if request.param == "swin":        # the synthetic integer buffer is the reference's own
        own = ref.state_dict()["video_encoder.layers.0.blocks.0.attn.relative_position_index"]
        assert torch.equal(own, synth.swin_relative_position_index(spec.swin_window))
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
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


def test_swin_shifted_block_with_depth_shift():
    """16-frame inputs shift windows along time too (window 8, shift 4): one reference block vs the restatement"""
    ref_harness._install()
    from model.videoswin import SwinTransformerBlock3D, compute_mask
    from valor_amd import synth
    from valor_oracle import Oracle
    torch.manual_seed(0)
    blk = SwinTransformerBlock3D(dim=64, num_heads=2, window_size=(8, 7, 7), shift_size=(4, 3, 3)).float().eval()
    x = torch.randn(2, 16, 14, 14, 64)
    m = compute_mask(16, 14, 14, (8, 7, 7), (4, 3, 3), x.device)
    assert torch.equal(m, Oracle.swin_shift_mask((16, 14, 14), (8, 7, 7), (4, 3, 3)))
    o = Oracle(synth.swin_spec(), {"b." + k: v for k, v in blk.state_dict().items()})
    with torch.no_grad():
        assert float((blk(x, m) - o.swin_block(x, "b.", 2, (8, 7, 7), (4, 3, 3), 0.0)).abs().max()) < 1e-5


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
    assert torch.allclose(r["feat_v"], o["feat_v"], atol=1e-5)


def test_text_only_mlm_plumbing_case(setup):
    """BASELINE configs[0] / SURVEY 8d config 1: text-only MLM on CPU, batch 2 -- the reference modules composed by hand
    (forward_pt cannot express it) against the oracle's restatement: loss and the gradients of every BERT / head tensor."""
    import torch.nn.functional as F
    spec, ref, orc, sd_o, batch = setup
    toks = batch["txt_tokens"]["bert_tokens"]
    for p in ref.parameters():
        p.grad = None
    for v in sd_o.values():
        if v.is_floating_point():
            v.grad = None
    random.seed(21)
    txt_input, txt_labels = ref.text_masker(toks, 0.15)
    o = ref.multimodal_encoder(txt_input, None, None, None, casual=False)
    loss_r = F.cross_entropy(ref.cls(o[txt_labels != -1]), txt_labels[txt_labels != -1])
    loss_r.backward()
    random.seed(21)
    loss_o = orc.text_mlm(toks)["mlm_loss"]
    loss_o.backward()
    assert abs(float(loss_r) - float(loss_o)) <= 2e-5 * abs(float(loss_r))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-5 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 150


def test_large_configuration_components():
    """BASELINE configs[3] ("VideoSwin-L + BERT-large") has no shipped reference config (modeling.py:578-587, 618-625 accept base
    models only): the reference CLASSES with the large hyper-parameters against the oracle's restatement at
    those widths -- SwinTransformer3D(embed 192, heads 6/12/24/48) on a 2-frame clip and a 2-layer slice of BertModel(1024 wide,
    16 heads, 4096 inner) with cross-attention over [video | audio] features, outputs and input / weight gradients."""
    ref_harness._install()
    from model.bert import BertConfig, BertModel
    from model.videoswin import SwinTransformer3D
    from valor_amd import synth
    from valor_oracle import Oracle
    torch.manual_seed(1)
    spec = synth.large_spec()
    # ---- VideoSwin-L (shallow stage 3 to keep the CPU time down: depths are a free hyper-parameter of both sides)
    depths = (1, 1, 2, 1)
    swin = SwinTransformer3D(embed_dim=192, depths=list(depths), num_heads=[6, 12, 24, 48], drop_path_rate=0.0).float().train()
    for p in swin.parameters():
        torch.nn.init.normal_(p, std=0.05)
    sd = {"video_encoder." + k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in swin.state_dict().items()}
    import dataclasses
    orc = Oracle(dataclasses.replace(spec, swin_depths=depths), sd)
    vid = torch.randn(1, 3, 2, 224, 224)
    y_ref = swin(vid).permute(0, 2, 3, 4, 1)
    y_orc = orc.swin_visual(vid)
    assert y_orc.shape == (1, 2, 49, 1536)
    assert torch.allclose(y_ref.reshape(y_orc.shape), y_orc, atol=2e-4, rtol=1e-4)
    y_ref.square().mean().backward(); y_orc.square().mean().backward()
    for k in ("layers.2.downsample.norm.weight", "layers.3.blocks.0.attn.relative_position_bias_table", "patch_embed.proj.weight"):
        gr, go = dict(swin.named_parameters())[k].grad, sd["video_encoder." + k].grad
        assert float((gr - go).norm()) <= 2e-3 * float(gr.norm()) + 1e-9, k
    assert dict(swin.named_parameters())["layers.2.downsample.norm.weight"].shape == (3072,)      # the wide LayerNorm row
    # ---- BERT-large decoder slice
    cfg = BertConfig.from_dict(dict(attention_probs_dropout_prob=0.0, hidden_act="gelu", hidden_dropout_prob=0.0, hidden_size=1024,
                                    initializer_range=0.02, intermediate_size=4096, max_position_embeddings=64, num_attention_heads=16,
                                    num_hidden_layers=2, type_vocab_size=2, vocab_size=500))
    cfg.checkpointing, cfg.has_cross_attn, cfg.cross_attn_type = False, True, "va_concate"
    bert = BertModel(cfg).float().train()
    bsd = {"multimodal_encoder." + k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in bert.state_dict().items()}
    borc = Oracle(dataclasses.replace(spec, layers=2, vocab=500, max_pos=64), bsd)
    toks = torch.tensor([[101, 7, 8, 9, 102, 0, 0, 0], [101, 11, 12, 102, 0, 0, 0, 0]])
    vfeat, afeat = torch.randn(2, 98, 1024), torch.randn(2, 129, 1024)
    for casual in (True, False):
        o_ref = bert(toks, None, vfeat, afeat, casual=casual)
        o_orc = borc.bert_model(toks, None, vfeat, afeat, casual)
        assert torch.allclose(o_ref, o_orc, atol=2e-4, rtol=1e-4), casual


def test_shipped_large_configuration_clip_l14_with_shared_bert():
    """config/pretrain-VALOR-large.json:10-15,70 -- the reference's real large configuration: clip_vit_large_14_336px video encoder at
    video_resolution 224 (width 1024, patch 14, 257 tokens; the 336-px positional embedding is resized at construction) +
    bert_base_uncased text encoder shared with the multimodal encoder, use_task_prompt, contra_loss_ratio 1.5, Contra_head linears
    to contra_dim 512, hidden_trans_video_multimodal, and its image-text task string `..._mlm%tv`. The reference derives the CLIP
    depth from the checkpoint keys and the BERT depth from its json, so the pin runs the true WIDTHS on 2-layer stacks: losses,
    every parameter gradient (the untouched CLIP text tower must get none), argmax ids."""
    import dataclasses
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec = dataclasses.replace(synth.clip_large_spec(), vis_layers=2, txt_layers=1, aud_layers=12, layers=2)
    ropts = ref_harness.default_opts(video_encoder_type="clip_vit_large_14_336px", txt_encoder_type="bert_base_uncased",
                                     use_task_prompt=True, contra_loss_ratio=1.5, video_resolution=224)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, clip_layers=(2, 1), bert_layers=2)
    sd = synth.make_state_dict(spec, seed=13)
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:4], unexpected[:4])
    assert [k for k, _, kind in synth.state_dict_layout(spec)] == list(ref.state_dict().keys())       # checkpoint layout incl. order
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab), use_task_prompt=True, contra_loss_ratio=1.5)
    batch = synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=32, seed=14)
    for task in ("pt_contra%tva%tv%ta_caption%tva%tv%ta", "pt_contra%tv_caption%tv_mlm%tv"):
        for p in ref.parameters():
            p.grad = None
        for v in sd_o.values():
            if v.is_floating_point():
                v.grad = None
        random.seed(5); r_out = ref(batch, task=task, compute_loss=True); sum(r_out.values()).backward()
        random.seed(5); o_out = orc.forward_pt(batch, task, compute_loss=True); sum(o_out.values()).backward()
        assert set(r_out) == set(o_out)
        for k in r_out:
            assert abs(float(r_out[k]) - float(o_out[k])) <= 2e-5 * abs(float(r_out[k])), (task, k, float(r_out[k]), float(o_out[k]))
        n = 0
        for name, p in ref.named_parameters():
            g = sd_o[name].grad
            if p.grad is None:
                assert g is None or float(g.abs().max()) == 0.0, name
                continue
            scale = max(float(p.grad.norm()), 1e-5 * p.grad.numel() ** 0.5)
            assert float((g - p.grad).norm()) / scale < 2e-4, (task, name)
            n += 1
        assert n > (250 if "tva" in task else 100)
        assert dict(ref.named_parameters())["clip_model.transformer.resblocks.0.attn.in_proj_weight"].grad is None      # CLIP text tower: unused
        assert dict(ref.named_parameters())["contra_head_v.linear.weight"].grad is not None
    with torch.no_grad():
        random.seed(6); r = ref(batch, task="pt_contra%tv_caption%tv_mlm%tv", compute_loss=False)
        random.seed(6); o = orc.forward_pt(batch, "pt_contra%tv_caption%tv_mlm%tv", compute_loss=False)
    for k in r:
        if "scores" in k:
            assert torch.equal(r[k].argmax(-1), o[k].argmax(-1)), k
    assert torch.allclose(r["feat_v"], o["feat_v"], atol=1e-5) and torch.allclose(r["feat_t"], o["feat_t"], atol=1e-5)


@pytest.mark.parametrize("res,frames", [(96, 3), (160, 2), (112, 10), (104, 10)])
def test_swin_padding_cases(res, frames):
    """feature maps that are not multiples of the (7, 7) window and odd PatchMerging inputs (videoswin.py:199-203, 222-223,
    257-259): 96 px -> 24 / 12 / 6 / 3, 160 px -> 40 / 20 / 10 / 5, 104 px -> 26 / 13 / 7 / 4 (odd maps into PatchMerging); 10 frames pad
    the depth 10 -> 16 against the 8-deep window, with a depth shift (the reference's finetune scripts test with 10 and 12 frames).
    Reference SwinTransformer3D vs the oracle, output and gradients. (tests/test_swin_gpu.py runs the native encoder on the same cases.)"""
    ref_harness._install()
    import dataclasses
    from model.videoswin import SwinTransformer3D
    from valor_amd import synth
    from valor_oracle import Oracle
    torch.manual_seed(2)
    depths = (2, 2, 2, 2)
    swin = SwinTransformer3D(embed_dim=32, depths=list(depths), num_heads=[1, 2, 4, 8], drop_path_rate=0.0).float().train()
    for p in swin.parameters():
        torch.nn.init.normal_(p, std=0.1)
    sd = {"video_encoder." + k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in swin.state_dict().items()}
    spec = dataclasses.replace(synth.swin_spec(), swin_embed=32, swin_depths=depths, swin_heads=(1, 2, 4, 8), resolution=res)
    orc = Oracle(spec, sd)
    vid = torch.randn(2, 3, frames, res, res)
    y_ref = swin(vid).permute(0, 2, 3, 4, 1)
    y_orc = orc.swin_visual(vid)
    assert torch.allclose(y_ref.reshape(y_orc.shape), y_orc, atol=1e-4, rtol=1e-4)
    y_ref.square().mean().backward(); y_orc.square().mean().backward()
    for k, p in swin.named_parameters():
        go = sd["video_encoder." + k].grad
        assert float((p.grad - go).norm()) <= 2e-3 * float(p.grad.norm()) + 1e-8, k


@pytest.mark.parametrize("variant", ["clip", "swin"])
def test_use_task_prompt_matches_reference(variant):
    """use_task_prompt=True (config/pretrain-VALOR-large.json:14): the caption passes get the 'describe the video ...' prompt
    (pretrain.py:436-439) and, with the BERT text encoder, the contrastive text pass gets 'project language in common space' whose
    rows are dropped again (pretrain.py:254-263): losses and all gradients against the reference."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    if variant == "swin":
        spec = synth.swin_spec()
        ropts = ref_harness.default_opts(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased", use_task_prompt=True)
    else:
        spec, ropts = synth.base_spec(), ref_harness.default_opts(use_task_prompt=True)
    sd = synth.make_state_dict(spec, seed=13)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0)
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab), use_task_prompt=True)
    batch = synth.make_batch(spec, batch=2, frames=1, audio_slices=1, txt_len=32, seed=14)
    random.seed(3); r = ref(batch, task=TASK, compute_loss=True); sum(r.values()).backward()
    random.seed(3); o = orc.forward_pt(batch, TASK, compute_loss=True); sum(o.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        assert abs(float(r[k]) - float(o[k])) <= 2e-5 * abs(float(r[k])), (k, float(r[k]), float(o[k]))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)      # analytically zero gradients are 1e-9 noise on both sides
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 800


def test_finetune_tasks_and_caption_generation(setup):
    """'ret%..' / 'cap%..' losses (config/fast-retrieval-*.json, caption-*.json) and generate_cap with greedy and beam
    decoding (model/pretrain.py:544-725,914-1189): the reference's sequences, token for token."""
    spec, ref, orc, sd_o, batch = setup
    with torch.no_grad():
        random.seed(3)
        r = ref(batch, task="ret%tva%tv", compute_loss=True)
        random.seed(3)
        o = orc.forward(batch, "ret%tva%tv", compute_loss=True)
        assert abs(float(r["contra_loss"]) - float(o["contra_loss"])) <= 2e-5 * abs(float(r["contra_loss"]))
        re, oe = ref(batch, task="ret%tva%tv", compute_loss=False), orc.forward(batch, "ret%tva%tv", compute_loss=False)
        for k in ("feat_t", "feat_v", "feat_a"):
            assert torch.allclose(re[k], oe[k], atol=1e-5), k
        random.seed(4)
        r = ref(dict(batch), task="cap%tva%tv", compute_loss=True)       # forward_cap replaces batch['txt_tokens'] in place
        random.seed(4)
        o = orc.forward(batch, "cap%tva%tv", compute_loss=True)
        assert abs(float(r["caption_loss"]) - float(o["caption_loss"])) <= 2e-5 * abs(float(r["caption_loss"]))
        old = ref.beam_size, ref.max_generation_len
        try:
            ref.max_generation_len = 6
            ref.beam_size = 1
            rg = ref(dict(batch), task="cap%tva%ta", compute_loss=False)
            og = orc.forward_cap(batch, "cap%tva%ta", compute_loss=False, beam_size=1, max_generation_len=6)
            for k in ("generated_sequences_t_va", "generated_sequences_t_a"):
                assert torch.equal(rg[k], og[k]), (k, rg[k], og[k])
            ref.beam_size = 3
            rb = ref(dict(batch), task="cap%tva%tv", compute_loss=False)
            ob = orc.forward_cap(batch, "cap%tva%tv", compute_loss=False, beam_size=3, max_generation_len=6)
            for k in ("generated_sequences_t_va", "generated_sequences_t_v"):
                assert torch.equal(rb[k], ob[k]), (k, rb[k], ob[k])
        finally:
            ref.beam_size, ref.max_generation_len = old



@pytest.mark.parametrize("prompt", [False, True])
def test_video_qa_task(prompt):
    """'qa%tva%tv' (config/VQA-msrvtt.json): forward_qa_single's per-sample-normalised loss with the question as the prompt rows, and
    generate_qa's greedy answers (beam_size_qa = 1, train_utils.py:693) -- model/pretrain.py:1191-1459, one answer per question."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec = synth.base_spec()
    ropts = ref_harness.default_opts(use_task_prompt=prompt)
    sd = synth.make_state_dict(spec, seed=50)
    ref = ref_harness.build_reference(ropts, state_dict=sd, dropout=0.0)
    orc = Oracle(spec, trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab), use_task_prompt=prompt)
    batch = synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=8, seed=51, questions=True)
    with torch.no_grad():
        random.seed(5)
        r = ref({k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}, task="qa%tva%tv", compute_loss=True)
        random.seed(5)
        o = orc.forward(batch, "qa%tva%tv", compute_loss=True)
        assert abs(float(r["qa_loss"]) - float(o["qa_loss"])) <= 2e-5 * abs(float(r["qa_loss"])), (float(r["qa_loss"]), float(o["qa_loss"]))
        ref.max_generation_len = 5
        rg = ref({k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}, task="qa%tva%ta", compute_loss=False)
        og = orc.forward_qa(batch, "qa%tva%ta", compute_loss=False, max_generation_len=5)
        for k in ("generated_answers_t_va", "generated_answers_t_a"):
            assert torch.equal(rg[k], og[k]), (k, rg[k], og[k])


def test_image_qa_with_several_weighted_answers():
    """forward_qa_single's tile_feats branch (model/pretrain.py:1243-1265,1288-1290; data/vqa.py:181-189): answer_nums = [2, 1, 3] ->
    six answer rows, the question / video / audio rows tiled per answer, loss rows weighted and summed over the three questions."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec = synth.base_spec()
    sd = synth.make_state_dict(spec, seed=50)
    ref = ref_harness.build_reference(None, state_dict=sd, dropout=0.0)
    orc = Oracle(spec, trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab))
    batch = synth.make_batch(spec, batch=3, frames=1, audio_slices=1, txt_len=8, seed=52, questions=True, answers_per_question=[2, 1, 3])
    with torch.no_grad():
        random.seed(6)
        r = ref({k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}, task="qa%tva%tv", compute_loss=True)
        random.seed(6)
        o = orc.forward(batch, "qa%tva%tv", compute_loss=True)
    assert abs(float(r["qa_loss"]) - float(o["qa_loss"])) <= 2e-5 * abs(float(r["qa_loss"])), (float(r["qa_loss"]), float(o["qa_loss"]))


def test_contrastive_groups_without_text_on_the_query_side(setup):
    """'va' / 'vta' / 'atv' (model/pretrain.py:346-370, forward_ret :653-680): video-audio alignment, and video / audio queries against the
    concatenated text + other-modality tokens -- retrieval finetune and pretraining grammar."""
    spec, ref, orc, sd_o, batch = setup
    with torch.no_grad():
        for task in ("ret%va", "ret%vta%atv", "pt_contra%tva%va%atv"):
            r = ref(batch, task=task, compute_loss=True)
            o = orc.forward(batch, task, compute_loss=True)
            assert abs(float(r["contra_loss"]) - float(o["contra_loss"])) <= 2e-5 * abs(float(r["contra_loss"])), (task, float(r["contra_loss"]), float(o["contra_loss"]))


def _shallow(spec, **kw):
    """the option-branch pins below do not depend on depth: 2 ViT / 1 CLIP-text / 2 BERT layers (the reference derives the CLIP depth from
    the checkpoint keys and the BERT depth from its json, ref_harness.build_reference) -- a third of the build and step time"""
    import dataclasses
    return dataclasses.replace(spec, vis_layers=2, txt_layers=1, layers=2, **kw)


SHALLOW = dict(clip_layers=(2, 1), bert_layers=2)


def test_caption_type_lm_matches_reference():
    """caption_type='lm' (model/pretrain.py:429-433, :812-816, :1230-1234): the caption passes read the unmasked tokens under the causal
    mask and predict the NEXT token at every position (padding / last position ignored) -- pretraining task string and the caption
    finetune loss against the unmodified reference: losses and every gradient."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec, ropts = _shallow(synth.base_spec()), ref_harness.default_opts(caption_type="lm")
    sd = synth.make_state_dict(spec, seed=17)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, **(SHALLOW if spec.video_encoder == 'clip' else dict(bert_layers=2)))
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab), caption_type="lm")
    batch = synth.make_batch(spec, batch=2, frames=1, audio_slices=1, txt_len=32, seed=18)
    random.seed(5); r = ref(batch, task=TASK, compute_loss=True); sum(r.values()).backward()
    random.seed(5); o = orc.forward_pt(batch, TASK, compute_loss=True); sum(o.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        assert abs(float(r[k]) - float(o[k])) <= 2e-5 * abs(float(r[k])), (k, float(r[k]), float(o[k]))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 150
    with torch.no_grad():
        rc = ref(dict(batch), task="cap%tva%tv", compute_loss=True)      # forward_cap replaces batch['txt_tokens'] in place
        oc = orc.forward(batch, "cap%tva%tv", compute_loss=True)
    assert abs(float(rc["caption_loss"]) - float(oc["caption_loss"])) <= 2e-5 * abs(float(rc["caption_loss"]))


def test_several_questions_per_clip_at_generation():
    """generate_qa with sample_num = [2, 1] (model/pretrain.py:1378-1390): three question rows over two clips, the clip rows expanded per
    question -- greedy answers of the reference, token for token."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec = _shallow(synth.base_spec())
    sd = synth.make_state_dict(spec, seed=50)
    ref = ref_harness.build_reference(ref_harness.default_opts(), state_dict=sd, dropout=0.0, **SHALLOW)
    orc = Oracle(spec, trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab))
    batch = synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=8, seed=53, questions=True)
    three = synth.make_batch(spec, batch=3, frames=1, audio_slices=1, txt_len=8, seed=54, questions=True)
    batch = dict(batch, question_tokens=three["question_tokens"], sample_num=[2, 1])
    with torch.no_grad():
        ref.max_generation_len = 5
        rg = ref({k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}, task="qa%tva%tv", compute_loss=False)
        og = orc.forward_qa(batch, "qa%tva%tv", compute_loss=False, max_generation_len=5)
    for k in ("generated_answers_t_va", "generated_answers_t_v"):
        assert rg[k].shape[0] == 3 and torch.equal(rg[k], og[k]), (k, rg[k], og[k])


def test_label_smoothing_of_the_caption_finetune_loss():
    """config.label_smoothing = 0.1 (model/pretrain.py:72-74, LabelSmoothing :46-61, used by forward_cap_single :839-840 only): the KL
    divergence to the smoothed target, against the unmodified reference -- loss and every gradient of 'cap%tva%tv'."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec, ropts = _shallow(synth.base_spec()), ref_harness.default_opts(label_smoothing=0.1)
    sd = synth.make_state_dict(spec, seed=19)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, **(SHALLOW if spec.video_encoder == 'clip' else dict(bert_layers=2)))
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab), label_smoothing=0.1)
    batch = synth.make_batch(spec, batch=2, frames=1, audio_slices=1, txt_len=32, seed=20)
    random.seed(7); r = ref(dict(batch), task="cap%tva%tv", compute_loss=True); r["caption_loss"].backward()
    random.seed(7); o = orc.forward(batch, "cap%tva%tv", compute_loss=True); o["caption_loss"].backward()
    assert abs(float(r["caption_loss"]) - float(o["caption_loss"])) <= 2e-5 * abs(float(r["caption_loss"])), (float(r["caption_loss"]), float(o["caption_loss"]))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 100


def test_full_masker_matches_reference():
    """config.full_masker (model/pretrain.py:79,137-142; bert.py:197-201,872-878): caption rows = [tokens | as many [MASK]s], [MASK] i at
    position i + 1 predicting token i + 1 under the block attention mask. forward_cap_single and forward_qa_single pass the flag to the
    decoder; forward_pt with it raises an IndexError in the reference (:454: 'tv' outputs sliced with the original length against doubled
    labels) and is refused here. Caption and QA finetune losses and every gradient of the caption loss, against the unmodified reference."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec, ropts = _shallow(synth.base_spec()), ref_harness.default_opts(full_masker=True)
    sd = synth.make_state_dict(spec, seed=21)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, **(SHALLOW if spec.video_encoder == 'clip' else dict(bert_layers=2)))
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab), full_masker=True)
    batch = synth.make_batch(spec, batch=2, frames=1, audio_slices=1, txt_len=16, seed=22)
    random.seed(9); r = ref(dict(batch), task="cap%tva%tv", compute_loss=True); r["caption_loss"].backward()
    random.seed(9); o = orc.forward(batch, "cap%tva%tv", compute_loss=True); o["caption_loss"].backward()
    assert abs(float(r["caption_loss"]) - float(o["caption_loss"])) <= 2e-5 * abs(float(r["caption_loss"])), (float(r["caption_loss"]), float(o["caption_loss"]))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 100
    with torch.no_grad():
        with pytest.raises(IndexError):          # forward_pt + full_masker: the reference slices 'tv' outputs with the original length (:454)
            ref(batch, task=TASK, compute_loss=True)
        with pytest.raises(NotImplementedError):
            orc.forward_pt(batch, TASK, compute_loss=True)
        qb = synth.make_batch(spec, batch=2, frames=1, audio_slices=1, txt_len=8, seed=23, questions=True)
        rq = ref({k: (dict(v) if isinstance(v, dict) else v) for k, v in qb.items()}, task="qa%tva%tv", compute_loss=True)
        oq = orc.forward(qb, "qa%tva%tv", compute_loss=True)
        assert abs(float(rq["qa_loss"]) - float(oq["qa_loss"])) <= 2e-5 * abs(float(rq["qa_loss"])), (float(rq["qa_loss"]), float(oq["qa_loss"]))


@pytest.mark.parametrize("opt", ["lm", "full_masker"])
def test_generation_with_lm_and_full_masker(opt):
    """generate_cap with caption_type='lm' ([CLS] + the tokens so far, the last token's logits: pretrain.py:1038-1040) and with full_masker
    (the generation branch of forward_cap_single, :878-900, never sees the flag: same sequences as plain 'unimlm' decoding) -- greedy and
    beam-3 sequences of the unmodified reference, token for token; video QA answers likewise."""
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    kw = dict(caption_type="lm") if opt == "lm" else dict(full_masker=True)
    spec, ropts = _shallow(synth.base_spec()), ref_harness.default_opts(**kw)
    sd = synth.make_state_dict(spec, seed=50)
    ref = ref_harness.build_reference(ropts, state_dict=sd, dropout=0.0, **SHALLOW)
    orc = Oracle(spec, trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab), **kw)
    batch = synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=8, seed=55, questions=True)
    cp = lambda: {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
    with torch.no_grad():
        old = ref.beam_size, ref.max_generation_len
        try:
            ref.max_generation_len = 5
            ref.beam_size = 1
            rg = ref(cp(), task="cap%tva%ta", compute_loss=False)
            og = orc.forward_cap(batch, "cap%tva%ta", compute_loss=False, beam_size=1, max_generation_len=5)
            for k in ("generated_sequences_t_va", "generated_sequences_t_a"):
                assert torch.equal(rg[k], og[k]), (k, rg[k], og[k])
            ref.beam_size = 3
            rb = ref(cp(), task="cap%tva%tv", compute_loss=False)
            ob = orc.forward_cap(batch, "cap%tva%tv", compute_loss=False, beam_size=3, max_generation_len=5)
            for k in ("generated_sequences_t_va", "generated_sequences_t_v"):
                assert torch.equal(rb[k], ob[k]), (k, rb[k], ob[k])
            rq = ref(cp(), task="qa%tva%tv", compute_loss=False)
            oq = orc.forward_qa(batch, "qa%tva%tv", compute_loss=False, max_generation_len=5)
            for k in ("generated_answers_t_va", "generated_answers_t_v"):
                assert torch.equal(rq[k], oq[k]), (k, rq[k], oq[k])
        finally:
            ref.beam_size, ref.max_generation_len = old


@pytest.mark.parametrize("variant,late", [("clip", False), ("clip", True), ("swin", False)])
def test_coarse_contrastive_matches_reference(variant, late):
    """contra_type='coarse' (model/pretrain.py:100-101,375-395; modeling.py:373-407): one pooled vector per modality, plain similarity
    matrices, the tva group through va_fusion (or, late_fusion, as the sum of the tv and ta matrices). Losses and every gradient of the
    pretraining task string against the unmodified reference; the parameter set is the reference's (no fine-weight heads, va_fusion)."""
    import dataclasses
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    if variant == "swin":
        spec = dataclasses.replace(synth.swin_spec(), contra_type="coarse", late_fusion=late, layers=2)
        ropts = ref_harness.default_opts(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased", contra_type="coarse", late_fusion=late)
    else:
        spec = _shallow(synth.base_spec(), contra_type="coarse", late_fusion=late)
        ropts = ref_harness.default_opts(contra_type="coarse", late_fusion=late)
    sd = synth.make_state_dict(spec, seed=31)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, **(SHALLOW if spec.video_encoder == 'clip' else dict(bert_layers=2)))
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    assert ("va_fusion.weight" in sd) == (not late) and "text_fine_weight.0.weight" not in sd
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    batch = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=32)
    random.seed(3); r = ref(batch, task=TASK, compute_loss=True); sum(r.values()).backward()
    random.seed(3); o = orc.forward_pt(batch, TASK, compute_loss=True); sum(o.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        assert abs(float(r[k]) - float(o[k])) <= 2e-5 * abs(float(r[k])), (k, float(r[k]), float(o[k]))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 150


def test_late_fusion_with_the_fine_matrix_matches_reference():
    """late_fusion with contra_type='fine' (model/pretrain.py:313-321): the tva group scores fine(t, v) + fine(t, a) with UNIT token weights
    (the fine-weight heads still serve tv / ta): losses and every gradient against the unmodified reference."""
    import dataclasses
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec = _shallow(synth.base_spec(), late_fusion=True)
    ropts = ref_harness.default_opts(late_fusion=True)
    sd = synth.make_state_dict(spec, seed=41)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, **(SHALLOW if spec.video_encoder == 'clip' else dict(bert_layers=2)))
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    batch = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=42)
    task = "pt_contra%tva%tv%ta"
    random.seed(3); r = ref(batch, task=task, compute_loss=True); r["contra_loss"].backward()
    random.seed(3); o = orc.forward_pt(batch, task, compute_loss=True); o["contra_loss"].backward()
    assert abs(float(r["contra_loss"]) - float(o["contra_loss"])) <= 2e-5 * abs(float(r["contra_loss"])), (float(r["contra_loss"]), float(o["contra_loss"]))
    n = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
    assert n > 100


@pytest.mark.parametrize("variant", ["clip", "swin"])
def test_frozen_options_freeze_the_reference_s_parameters(variant):
    """frozen_vision / frozen_multimodal (model/modeling.py:319-322, 675-680): the native model must clear requires_grad on exactly the
    parameters the UNMODIFIED reference constructor clears (packed q|k|v / k|v slots: all of their reference tensors agree)."""
    from valor_amd import synth
    from valor_amd.model.valor import VALOR
    if variant == "swin":
        spec = synth.swin_spec()
        kw = dict(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased")
    else:
        spec, kw = synth.base_spec(), {}
    ref = ref_harness.build_reference(ref_harness.default_opts(frozen_vision=True, frozen_multimodal=True, **kw), state_dict=None, dropout=0.0)
    ref_frozen = {n for n, p in ref.named_parameters() if not p.requires_grad}
    assert ref_frozen                                         # (VideoSwin: frozen_vision has no branch in the reference; the decoder rules apply)
    model = VALOR({"frozen_vision": True, "frozen_multimodal": True, "dropout": 0.0}, spec=spec, dtype=torch.float32, device="cpu")
    ours = set()
    for name, _shape, refs in model.table:
        flags = {r in ref_frozen or (r == "cls.decoder.weight" and "multimodal_encoder.embeddings.word_embeddings.weight" in ref_frozen) for r in refs}
        assert len(flags) == 1, (name, refs)                  # a packed slot is frozen as a whole or not at all
        if not model.P[name].requires_grad:
            ours.update(refs)
    known = {r for _n, _s, refs in model.table for r in refs}
    tied = {"cls.decoder.weight"} & known                     # the reference lists the tied matrix once (under the embedding's name)
    assert ours - tied == ref_frozen & known, (sorted((ours - tied) ^ (ref_frozen & known))[:10])
    # nothing frozen by default
    plain = VALOR({"dropout": 0.0}, spec=spec, dtype=torch.float32, device="cpu")
    assert all(p.requires_grad for p in plain.P.values())


@pytest.mark.parametrize("mode", ["va_parallel", "video_audio", "audio_video"])
def test_cross_attention_block_per_modality_matches_reference(mode):
    """cross_attn_type != 'va_concate' (model/bert.py:430-436,459-496): a cross-attention block per modality (cross_attn_v / cross_attn_a with
    their own query / key / value / output / LayerNorm), summed or applied one after the other; a group with one modality runs that
    modality's block alone. Pretraining task string (groups tva, tv, ta) against the unmodified reference: losses and every gradient --
    incl. those of both blocks' parameters --, the caption finetune loss, greedy captions."""
    import dataclasses
    from valor_amd import synth
    from valor_oracle import Oracle, trainable_copy
    spec, ropts = dataclasses.replace(_shallow(synth.base_spec()), cross_attn_type=mode), ref_harness.default_opts(cross_attn_type=mode)
    sd = synth.make_state_dict(spec, seed=27)
    ref = ref_harness.build_reference(ropts, state_dict=None, dropout=0.0, **SHALLOW)
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    sd_o = trainable_copy(sd)
    orc = Oracle(spec, sd_o, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    batch = synth.make_batch(spec, batch=2, frames=1, audio_slices=1, txt_len=32, seed=28)
    random.seed(5); r = ref(batch, task=TASK, compute_loss=True); sum(r.values()).backward()
    random.seed(5); o = orc.forward_pt(batch, TASK, compute_loss=True); sum(o.values()).backward()
    for k in ("contra_loss", "caption_loss", "mlm_loss"):
        assert abs(float(r[k]) - float(o[k])) <= 2e-5 * abs(float(r[k])), (k, float(r[k]), float(o[k]))
    n = blocks = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = sd_o[name].grad
        scale = max(float(p.grad.norm()), 1e-4 * p.grad.numel() ** 0.5)
        assert float((g - p.grad).norm()) / scale < 2e-4, name
        n += 1
        blocks += ("cross_attn_v." in name) or ("cross_attn_a." in name)
    assert n > 150 and blocks == 2 * spec.layers * 10          # query / key / value / dense / LayerNorm (weight + bias) of both blocks of every layer
    with torch.no_grad():
        random.seed(6); rc = ref(dict(batch), task="cap%tva%tv%ta", compute_loss=True)
        random.seed(6); oc = orc.forward(batch, "cap%tva%tv%ta", compute_loss=True)
        assert abs(float(rc["caption_loss"]) - float(oc["caption_loss"])) <= 2e-5 * abs(float(rc["caption_loss"]))
        ref.max_generation_len, ref.beam_size = 5, 1
        rg = ref(dict(batch), task="cap%tva%tv", compute_loss=False)
        og = orc.forward_cap(batch, "cap%tva%tv", compute_loss=False, beam_size=1, max_generation_len=5)
        for k in ("generated_sequences_t_va", "generated_sequences_t_v"):
            assert torch.equal(rg[k], og[k]), k
