"""TEST INFRASTRUCTURE: generate tests/golden/*.pt by running the UNMODIFIED reference (/root/reference)
on CPU fp32 with seeded synthetic weights/inputs (valor_amd/synth.py), dropout 0 (and VideoSwin drop-path 0). Run in the build
container only:  python oracle/make_goldens.py
Each fixture holds inputs' recipe (spec, seeds, shapes), the masked token tensors the reference's
TokenMasker produced, losses, score matrices, argmax ids, activation slices, per-parameter gradient
norms + a few gradient slices, and the parameters after 2 reference optimizer steps
(optim/misc.py build_optimizer + optim/adamw.py AdamW + clip_grad_norm_ 5.0 + warmup_linear)."""
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402
from valor_amd import synth  # noqa: E402

TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"
SLICE_KEYS = ["clip_model.visual.transformer.resblocks.0.attn.in_proj_weight", "clip_model.visual.conv1.weight",
              "audio_encoder.layer.11.ff_layer.linear2.weight", "multimodal_encoder.encoder.layer.0.cross_attn.cross.key.weight",
              "multimodal_encoder.embeddings.word_embeddings.weight", "cls.decoder.bias", "clip_model.logit_scale",
              "text_fine_weight.0.weight", "video_frame_embedding", "clip_model.token_embedding.weight"]


SWIN_SLICE_KEYS = ["video_encoder.patch_embed.proj.weight", "video_encoder.layers.0.blocks.1.attn.relative_position_bias_table",
                   "video_encoder.layers.2.blocks.17.attn.qkv.weight", "video_encoder.layers.1.downsample.reduction.weight",
                   "video_encoder.layers.3.blocks.1.mlp.fc2.weight", "hidden_trans_video_multimodal.0.weight",
                   "contra_head_v.linear.weight", "contra_head_t.linear.weight", "contra_temp",
                   "audio_encoder.layer.11.ff_layer.linear2.weight", "multimodal_encoder.encoder.layer.0.cross_attn.cross.key.weight",
                   "multimodal_encoder.embeddings.word_embeddings.weight", "cls.decoder.bias", "video_frame_embedding"]


def run(name, batch_size, frames, audio_slices, wseed, bseed, mseed, variant="clip", task=None, bf16_exact=False, steps=2, ref_opts=None):
    global SLICE_KEYS
    TASK = task or globals()["TASK"]
    build_kw, model_opts = {}, {}
    if variant == "swin":                  # scripts/pretrain.sh:3-8
        spec = synth.swin_spec()
        ropts = ref_harness.default_opts(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased")
        SLICE_KEYS = SWIN_SLICE_KEYS
    elif variant == "clip_large":
        # config/pretrain-VALOR-large.json:10-15 -- the reference's shipped large configuration (CLIP ViT-L/14 at 224 px + shared
        # bert_base_uncased, use_task_prompt, contra_loss_ratio 1.5) at its true WIDTHS (1024-wide ViT, patch 14, 257 tokens per frame,
        # video 1024 -> hidden 768 through hidden_trans_video_multimodal) on 2-layer stacks: the reference derives the CLIP depth from
        # the checkpoint keys and the BERT depth from its json, so the unmodified code runs at the same depth as the native model
        import dataclasses
        spec = dataclasses.replace(synth.clip_large_spec(), vis_layers=2, txt_layers=1, aud_layers=12, layers=2)
        model_opts = dict(use_task_prompt=True, contra_loss_ratio=1.5)
        ropts = ref_harness.default_opts(video_encoder_type="clip_vit_large_14_336px", txt_encoder_type="bert_base_uncased",
                                         video_resolution=224, **model_opts)
        build_kw = dict(clip_layers=(2, 1), bert_layers=2)
        SLICE_KEYS = ["clip_model.visual.transformer.resblocks.0.attn.in_proj_weight", "clip_model.visual.conv1.weight",
                      "hidden_trans_video_multimodal.0.weight", "contra_head_v.linear.weight", "contra_head_t.linear.weight", "contra_temp",
                      "audio_encoder.layer.11.ff_layer.linear2.weight", "multimodal_encoder.encoder.layer.0.cross_attn.cross.key.weight",
                      "multimodal_encoder.embeddings.word_embeddings.weight", "cls.decoder.bias", "video_frame_embedding"]
    else:
        # ref_opts: options of the reference's own constructor (e.g. its `checkpointing` flag, modeling.py:573-627: same arithmetic, the
        # layers' activations recomputed in backward -- what lets the B = 64 training pass fit the build container's memory)
        spec, ropts = synth.base_spec(), (ref_harness.default_opts(**ref_opts) if ref_opts else None)
    sd = synth.make_state_dict(spec, seed=wseed, bf16_exact=bf16_exact)
    ref = ref_harness.build_reference(ropts, state_dict=sd, dropout=0.0, **build_kw)
    batch = synth.make_batch(spec, batch=batch_size, frames=frames, audio_slices=audio_slices, txt_len=32, seed=bseed, bf16_exact=bf16_exact)
    g = {"recipe": dict(spec=spec.to_dict(), weight_seed=wseed, batch_seed=bseed, masker_seed=mseed, batch=batch_size,
                        frames=frames, audio_slices=audio_slices, txt_len=32, task=TASK, bf16_exact=bf16_exact, model_opts=model_opts,
                        ref_opts=dict(ref_opts or {}))}
    # ---- eval pass (compute_loss=False): argmax ids + features
    with torch.no_grad():
        random.seed(mseed)
        ev = ref(batch, task=TASK, compute_loss=False)
    g["eval"] = {k: ev[k].argmax(-1) for k in ev if "scores" in k}
    g["eval"]["top2_margin_min"] = {k: float((ev[k].topk(2, -1).values[:, 0] - ev[k].topk(2, -1).values[:, 1]).min()) for k in ev if "scores" in k}
    # per-row top-1 / top-2 logit gap of the fp32 reference: a lower-precision run can only be held to the argmax on rows whose
    # gap exceeds its logit error
    g["eval"]["top2_margin"] = {k: (ev[k].topk(2, -1).values[:, 0] - ev[k].topk(2, -1).values[:, 1]).clone() for k in ev if "scores" in k}
    g["eval"].update(feat_t=ev["feat_t"], feat_v=ev["feat_v"], feat_a=ev["feat_a"], txt_labels_caption=ev["txt_labels_caption"],
                     txt_labels_mlm=ev["txt_labels_mlm"])
    # ---- training pass + 2 optimizer steps
    from easydict import EasyDict
    from optim.misc import build_optimizer
    from optim.sched import get_lr_sched
    opts = EasyDict(learning_rate=1e-4, weight_decay=0.01, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=0.0, decoder_lr=-1,
                    new_params_name=[], optim="adamw", betas=[0.9, 0.98], warmup_ratio=0.1, num_train_steps=10,
                    scheduler="warmup_linear", grad_norm=5.0)
    opt = build_optimizer(ref, opts)
    g["steps"] = []
    for step in range(steps):
        opt.zero_grad()
        random.seed(mseed + step)
        out = ref(batch, task=TASK, compute_loss=True)
        loss = sum(out.values())
        loss.backward()
        rec = {"losses": {k: float(v) for k, v in out.items()}}
        if step == 0:
            rec["grad_norm"] = {n: float(p.grad.norm()) for n, p in ref.named_parameters() if p.grad is not None}
            rec["no_grad"] = [n for n, p in ref.named_parameters() if p.grad is None]
            rec["grad_slices"] = {n: dict(ref.named_parameters())[n].grad.reshape(-1)[:64].clone() for n in SLICE_KEYS
                                  if dict(ref.named_parameters())[n].grad is not None}      # single-modality tasks leave an encoder unused
        lr_ratio = get_lr_sched(step + 1, opts)
        for pg in opt.param_groups:
            pg["lr"] = pg["init_lr"] * lr_ratio
        rec["lr_ratio"] = lr_ratio
        rec["total_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(ref.parameters(), opts.grad_norm))
        opt.step()
        g["steps"].append(rec)
    g["after_2_steps"] = None if steps < 2 else {"param_norm": {n: float(p.detach().double().norm()) for n, p in ref.named_parameters()},
                          "param_slices": {n: dict(ref.named_parameters())[n].detach().reshape(-1)[:64].clone() for n in SLICE_KEYS},
                          "delta_norm": {n: float((p.detach() - sd[n]).double().norm()) for n, p in ref.named_parameters()}}
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    path = os.path.join(ROOT, "tests", "golden", name + ".pt")
    torch.save(g, path)
    print(name, {k: round(v, 6) for k, v in g["steps"][0]["losses"].items()}, "->", path, os.path.getsize(path) // 1024, "KiB")


FIXTURES = {
    "ref_base_b2f2a1": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50),
    "ref_base_b3f1a2": dict(batch_size=3, frames=1, audio_slices=2, wseed=7, bseed=8, mseed=9),
    "ref_swin_b2f2a1": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50, variant="swin"),
    # single-modality tasks (datasets without audio / without video): the decoder cross-attends to one modality only
    "ref_base_b2f2a1_tv": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50, task="pt_contra%tv_caption%tv_mlm%tv"),
    "ref_base_b2f2a1_ta": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50, task="pt_contra%ta_caption%ta_mlm%ta"),
    # "_q": weights and inputs are bf16-representable fp32 values, so the bf16 native model and the fp32 reference run on
    # IDENTICAL tensors; b2f8a2 = the geometry bench.py times (8 frames, 2 audio slices: 1834 cross-attention keys, frame
    # embedding rows 0..7, three kv_range groups at F = 8)
    "ref_base_b2f2a1_q": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50, bf16_exact=True),
    "ref_base_b2f8a2_q": dict(batch_size=2, frames=8, audio_slices=2, wseed=21, bseed=22, mseed=23, bf16_exact=True),
    "ref_swin_b2f8a2_q": dict(batch_size=2, frames=8, audio_slices=2, wseed=21, bseed=22, mseed=23, bf16_exact=True, variant="swin"),
    # a batch of 16: the InfoNCE loss of a B x B score matrix averages 2B terms around ln B, so its RELATIVE sensitivity to feature noise
    # falls like 1 / (sqrt(B) ln B); B = 2 is the worst case by construction (profiles/r02_bf16_attribution_b2f2a1.json)
    "ref_base_b16f2a1_q": dict(batch_size=16, frames=2, audio_slices=1, wseed=31, bseed=32, mseed=33, bf16_exact=True, steps=1),
    # B = 16 at the bench geometry (8 frames, 2 audio slices): 25 216 ViT token rows = 1188 tiles of 256 x 256 in the fc1 forward, a
    # wgrad contraction of 25 216 and >= 1024-tile dgrads -- the model-level bf16 run crosses every threshold of the GEMM policy
    # (csrc/gemm.hip use_8ph), so the 8-phase NN / NT / TT kernels are compared with the reference inside the step they are timed in
    "ref_base_b16f8a2_q": dict(batch_size=16, frames=8, audio_slices=2, wseed=41, bseed=42, mseed=43, bf16_exact=True, steps=1),
    # the shipped LARGE configuration's widths (CLIP ViT-L/14, 257 tokens per frame, LayerNorm rows of 1024, a 588-deep patch GEMM,
    # task prompt rows in the caption passes) -- full width, two-layer stacks on both sides; B = 8: the contrastive tolerance argument above
    # BASELINE configs[4]'s clip length, 16 frames: frame-embedding rows 0..15 (modeling.py:485-493); 16 * 197 + 2 * 129 = 3410 (CLIP) and
    # 16 * 49 + 2 * 129 = 1042 (VideoSwin) cross-attention keys (bert.py:314-340,448-457); VideoSwin feature maps 16 deep (PatchEmbed3D
    # pads one frame and strides by 1 along time, videoswin.py:355,367): TWO (8,7,7) windows along time and the (4,3,3) shift
    # (videoswin.py:196-223), the shift-mask regions along time
    "ref_base_b2f16a2_q": dict(batch_size=2, frames=16, audio_slices=2, wseed=71, bseed=72, mseed=73, bf16_exact=True),
    "ref_swin_b2f16a2_q": dict(batch_size=2, frames=16, audio_slices=2, wseed=71, bseed=72, mseed=73, bf16_exact=True, variant="swin"),
    # the same clip length at B = 8 for the bf16 comparison: the B = 2 fixture's 2 x 2 InfoNCE matrix is the worst case for the RELATIVE
    # contrastive tolerance (see ref_base_b16f2a1_q above; bf16 measured 1.29e-3 on ref_base_b2f16a2_q), B = 8 is held to the 1e-3
    "ref_base_b8f16a2_q": dict(batch_size=8, frames=16, audio_slices=2, wseed=81, bseed=82, mseed=83, bf16_exact=True, steps=1),
    # THE configuration bench.py times (BASELINE configs[1]: B = 64, 8 frames, 2 audio slices): 100 864 ViT rows, the 64-sample kv_range
    # groups and the 8 832-row decoder stack, the 64 x 64 fused contrastive matrix. One training pass of the unmodified reference with
    # its own `checkpointing` option (fp32, ~90 GB of saved activations without it), no optimizer step
    "ref_base_b64f8a2_q": dict(batch_size=64, frames=8, audio_slices=2, wseed=91, bseed=92, mseed=93, bf16_exact=True, steps=1,
                               ref_opts=dict(checkpointing=True)),
    "ref_cliplarge_b8f2a1_q": dict(batch_size=8, frames=2, audio_slices=1, wseed=61, bseed=62, mseed=63, bf16_exact=True, steps=1, variant="clip_large"),
}


def run_finetune(name, batch_size, frames, audio_slices, wseed, bseed, mseed, variant="clip", max_len=12):
    """SURVEY 8f row 4 fixture: 'ret%tva%tv' / 'cap%tva%tv' losses (config/fast-retrieval-*.json, caption-*.json) and generate_cap with
    greedy and beam-3 decoding, all from the UNMODIFIED reference. Random weights never produce [SEP], so a second greedy run raises
    cls.decoder.bias[102] by `eos_bias_delta` (chosen from the reference's own max-logit - [SEP]-logit gaps) to make rows end early and
    exercise the unfinished / EOS-fill logic (pretrain.py:1017-1020). Beam search is left without EOS: finished beams tie (see decode.py)."""
    from valor_oracle import Oracle, trainable_copy
    if variant == "swin":
        spec = synth.swin_spec()
        ropts = ref_harness.default_opts(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased")
    else:
        spec, ropts = synth.base_spec(), None
    sd = synth.make_state_dict(spec, seed=wseed, bf16_exact=True)
    ref = ref_harness.build_reference(ropts, state_dict=sd, dropout=0.0)
    batch = synth.make_batch(spec, batch=batch_size, frames=frames, audio_slices=audio_slices, txt_len=32, seed=bseed, bf16_exact=True)
    g = {"recipe": dict(spec=spec.to_dict(), weight_seed=wseed, batch_seed=bseed, masker_seed=mseed, batch=batch_size, frames=frames,
                        audio_slices=audio_slices, txt_len=32, bf16_exact=True, max_generation_len=max_len, beam_size=3)}
    orc = Oracle(spec, trainable_copy(sd), vocab_tokens=synth.synthetic_vocab(spec.vocab))
    with torch.no_grad():
        random.seed(mseed)
        g["ret_loss"] = float(ref(batch, task="ret%tva%tv", compute_loss=True)["contra_loss"])
        ev = ref(batch, task="ret%tva%tv", compute_loss=False)
        g["ret_feats"] = {k: ev[k].clone() for k in ("feat_t", "feat_v", "feat_a")}
        random.seed(mseed)
        g["cap_loss"] = float(ref(dict(batch), task="cap%tva%tv", compute_loss=True)["caption_loss"])
        ref.max_generation_len = max_len
        ref.beam_size = 1
        gr = ref(dict(batch), task="cap%tva%tv%ta", compute_loss=False)
        g["greedy"] = {k: v.clone() for k, v in gr.items() if k.startswith("generated")}
        ref.beam_size = 3
        bm = ref(dict(batch), task="cap%tva%tv", compute_loss=False)
        g["beam3"] = {k: v.clone() for k, v in bm.items() if k.startswith("generated")}
        # margins, from the restatement (tests/test_oracle_vs_reference.py pins it on the reference token for token)
        vo, ao = orc.forward_video_encoder(batch["video_pixels"]), orc.forward_audio_encoder(batch["audio_spectrograms"])
        vi, ai = orc.multimodal_inputs(vo, ao, batch_size)
        trace, gaps = [], []
        s, _ = orc.decode_greedy(vi, ai, None, batch_size, max_len, trace)
        assert torch.equal(s, g["greedy"]["generated_sequences_t_va"])
        top = torch.stack([t.topk(2, -1).values for t in trace])                       # [steps, b, 2]
        g["greedy_margin_t_va"] = (top[..., 0] - top[..., 1]).t().clone()               # [b, steps]
        sb = orc.decode_beam(vi, ai, None, batch_size, 3, max_len, gaps=gaps)
        assert torch.equal(sb, g["beam3"]["generated_sequences_t_va"])
        g["beam3_gap_t_va"] = torch.stack(gaps, 1).clone()                              # [b, steps]
        # video QA (config/VQA-*.json 'qa%tva%tv'): per-sample-normalised loss with the question as prompt rows, greedy answers
        qb = synth.make_batch(spec, batch=batch_size, frames=frames, audio_slices=audio_slices, txt_len=32, seed=bseed, bf16_exact=True, questions=True)
        fresh = lambda: {k: (dict(v) if isinstance(v, dict) else v) for k, v in qb.items()}
        random.seed(mseed)
        g["qa_loss"] = float(ref(fresh(), task="qa%tva%tv", compute_loss=True)["qa_loss"])
        ref.beam_size_qa = 1
        qa = ref(fresh(), task="qa%tva%tv", compute_loss=False)
        g["qa_greedy"] = {k: v.clone() for k, v in qa.items() if k.startswith("generated")}
        qtrace = []
        sq, _ = orc.decode_greedy(vi, ai, orc.qa_prompt(qb["question_tokens"]["bert_tokens"]), batch_size, max_len, qtrace)
        assert torch.equal(sq, g["qa_greedy"]["generated_answers_t_va"])
        top = torch.stack([t.topk(2, -1).values for t in qtrace])
        g["qa_greedy_margin_t_va"] = (top[..., 0] - top[..., 1]).t().clone()
        eos_gap = torch.stack([t.max(-1).values - t[:, 102] for t in trace])            # [steps, b]
        # [SEP] wins at step t of row r iff eos_gap[t, r] < delta (the trajectory before that is the unbiased one): pick the delta that
        # ends the rows at different steps (not all at step 0) with the widest clearance to every gap met on the way
        best = None
        for cand in sorted(set(eos_gap.flatten().tolist())):
            d = cand + 0.02
            ends = []
            clear = 1e9
            for r in range(batch_size):
                hit = (eos_gap[:, r] < d).nonzero()
                e = int(hit[0]) if hit.numel() else max_len
                ends.append(e)
                clear = min(clear, float((eos_gap[:e + 1, r] - d).abs().min()))
            if max(ends) >= 1 and min(ends) < max_len - 1 and len(set(ends)) > 1 and (best is None or clear > best[0]):
                best = (clear, d, ends)
        assert best is not None and best[0] > 5e-3, best
        delta = best[1]
        g["recipe"]["eos_end_steps"] = best[2]
        g["recipe"]["eos_bias_delta"] = delta
        sd2 = dict(sd)
        sd2["cls.decoder.bias"] = sd["cls.decoder.bias"].clone()
        sd2["cls.decoder.bias"][102] += delta
        ref.load_state_dict(sd2, strict=False)
        ref.beam_size = 1
        ge = ref(dict(batch), task="cap%tva%tv%ta", compute_loss=False)
        g["greedy_eos"] = {k: v.clone() for k, v in ge.items() if k.startswith("generated")}
        orc2 = Oracle(spec, trainable_copy(sd2), vocab_tokens=synth.synthetic_vocab(spec.vocab))
        trace = []
        s, _ = orc2.decode_greedy(vi, ai, None, batch_size, max_len, trace)
        assert torch.equal(s, g["greedy_eos"]["generated_sequences_t_va"])
        top = torch.stack([t.topk(2, -1).values for t in trace])
        g["greedy_eos_margin_t_va"] = (top[..., 0] - top[..., 1]).t().clone()
    path = os.path.join(ROOT, "tests", "golden", name + ".pt")
    torch.save(g, path)
    print(name, "ret", round(g["ret_loss"], 5), "cap", round(g["cap_loss"], 5), "qa", round(g["qa_loss"], 5), float(g["qa_greedy_margin_t_va"].min()), "greedy", g["greedy"]["generated_sequences_t_va"][0, :6].tolist(),
          "eos run", g["greedy_eos"]["generated_sequences_t_va"].tolist(), "min margins", float(g["greedy_margin_t_va"].min()),
          float(g["beam3_gap_t_va"].min()), "->", path, os.path.getsize(path) // 1024, "KiB")


FINETUNE_FIXTURES = {
    "ref_base_b2f2a1_ft": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50),
    "ref_swin_b2f2a1_ft": dict(batch_size=2, frames=2, audio_slices=1, wseed=50, bseed=51, mseed=50, variant="swin"),
}


if __name__ == "__main__":
    assert ref_harness.available()
    for name in (sys.argv[1:] or list(FIXTURES) + list(FINETUNE_FIXTURES)):
        if name in FINETUNE_FIXTURES:
            run_finetune(name, **FINETUNE_FIXTURES[name])
        else:
            run(name, **FIXTURES[name])
