"""Synthetic, seeded inputs and random-init weights of the VALOR architecture (no network here for
checkpoints or datasets). The state dict uses the reference's key names and shapes
(probed from model/pretrain.py::VALOR.state_dict()), so the same tensors load
into the reference (strict), the oracle and valor_amd.
"""
from dataclasses import dataclass, asdict

import torch


@dataclass
class ValorSpec:
    """Architecture hyper-parameters (defaults = VALOR-base, CLIP-ViT-B/16 variant,
    config/pretrain-VALOR-base.json). head_dim is 64 everywhere (width = 64 * heads)."""
    # CLIP visual (model/clip.py:220-274)
    vis_width: int = 768
    vis_layers: int = 12
    patch: int = 16
    resolution: int = 224
    # CLIP text (model/clip.py:317-331)
    txt_width: int = 512
    txt_layers: int = 12
    ctx_len: int = 77
    clip_vocab: int = 49408
    embed_dim: int = 512          # CLIP joint dim == contra_dim; text heads = embed_dim // 64 (clip.py:508)
    # AST (model/modeling.py:270-278, 738-762)
    aud_width: int = 768
    aud_layers: int = 12
    aud_inter: int = 3072
    melbins: int = 64
    target_len: int = 512
    aud_patch: int = 16
    # BERT multimodal decoder (model/bert.py:70-81)
    hidden: int = 768
    layers: int = 12
    inter: int = 3072
    vocab: int = 30522
    max_pos: int = 512
    # variant (scripts/pretrain.sh:3-8): video "clip" (CLIP-ViT) | "swin" (VideoSwin, model/videoswin.py:378-458);
    # text "clip" (CLIP text tower) | "bert" (the shared multimodal BERT run without cross-attention, modeling.py:688-691)
    video_encoder: str = "clip"
    txt_encoder: str = "clip"
    # VideoSwin (model/modeling.py:585-587, model/videoswin.py:378-399); head_dim is 32 everywhere
    swin_embed: int = 128
    swin_depths: tuple = (2, 2, 18, 2)
    swin_heads: tuple = (4, 8, 16, 32)
    swin_window: tuple = (8, 7, 7)
    swin_drop_path: float = 0.2
    # width of the contrastive space when the heads are Contra_head linears (pretrain.py:93-97, opts.contra_dim = 512) and it
    # differs from CLIP's joint embedding dim (CLIP-L/14: 768); 0 -> embed_dim (every shipped base configuration)
    contra_dim: int = 0
    # contrastive flavour (pretrain.py:75,83,100-116): "fine" = token-level MGA matrix with the fine-weight heads (every shipped config);
    # "coarse" = one pooled vector per modality and plain similarity matrices, with a va_fusion Linear for the tva group unless late_fusion
    contra_type: str = "fine"
    late_fusion: bool = False
    # how a decoder layer attends to the video / audio tokens (model/bert.py:430-436,447-496): "va_concate" = ONE cross-attention block over
    # the concatenated [video | audio] tokens (every shipped configuration); "va_parallel" / "video_audio" / "audio_video" = a block per
    # modality (cross_attn_v, cross_attn_a: own query / key / value / output / LayerNorm), their outputs summed or applied one after the other
    cross_attn_type: str = "va_concate"

    @property
    def cross_blocks(self):
        """names of a decoder layer's cross-attention blocks"""
        return ("cross_attn",) if self.cross_attn_type == "va_concate" else ("cross_attn_v", "cross_attn_a")

    @property
    def cdim(self):
        return self.contra_dim or self.embed_dim

    @property
    def clip_heads(self):
        """pretrain.py:89-92: the CLIP projections ARE the contrastive heads only when both encoders are CLIP towers"""
        return self.video_encoder == "clip" and self.txt_encoder == "clip"

    @property
    def swin_out(self):
        return self.swin_embed * 2 ** (len(self.swin_depths) - 1)

    @property
    def video_dim(self):
        """modeling.py:314,587"""
        return self.swin_out if self.video_encoder == "swin" else self.vis_width

    @property
    def txt_dim(self):
        """modeling.py:691,718"""
        return self.hidden if self.txt_encoder == "bert" else self.txt_width

    @property
    def swin_table(self):
        wd, wh, ww = self.swin_window
        return (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1)

    @property
    def vis_heads(self):
        return self.vis_width // 64

    @property
    def txt_heads(self):
        return self.embed_dim // 64

    @property
    def aud_heads(self):
        return self.aud_width // 64

    @property
    def heads(self):
        return self.hidden // 64

    @property
    def vis_tokens(self):
        return (self.resolution // self.patch) ** 2 + 1

    @property
    def aud_tokens(self):
        return (self.melbins // self.aud_patch) * (self.target_len // self.aud_patch) + 1

    def to_dict(self):
        return asdict(self)


def base_spec():
    return ValorSpec()


def swin_spec():
    """scripts/pretrain.sh "VALOR-base": VideoSwin-B + BERT text (shared with the decoder) + AST + BERT decoder."""
    return ValorSpec(video_encoder="swin", txt_encoder="bert")


def tiny_swin_spec():
    """3-stage VideoSwin (28 -> 14 -> 7 at 112 px: no padding anywhere) on the tiny BERT/AST of tiny_spec()."""
    return ValorSpec(video_encoder="swin", txt_encoder="bert", resolution=112, swin_embed=64, swin_depths=(2, 2, 2),
                     swin_heads=(2, 4, 8), swin_drop_path=0.2, embed_dim=128, aud_width=128, aud_layers=2, aud_inter=256,
                     melbins=32, target_len=64, aud_patch=16, hidden=128, layers=2, inter=256, vocab=1200, max_pos=64)


def clip_large_spec():
    """config/pretrain-VALOR-large.json:10-14, the reference's shipped LARGE configuration: clip_vit_large_14_336px video encoder
    run at video_resolution 224 (ViT-L/14: width 1024, 24 layers, 16 heads, 16 x 16 + 1 = 257 tokens per frame; its positional
    embedding is resized from the 336-px grid at construction, clip.py:481-491) + bert_base_uncased text encoder shared with the
    multimodal encoder. The whole CLIP model is constructed (modeling.py:560-573), so its text tower (width 768, 12 layers) and the
    two CLIP projections are parameters of the checkpoint although forward_pt never touches them; the contrastive heads are
    Contra_head linears to opts.contra_dim = 512 (pretrain.py:93-97); video 1024 -> hidden 768 goes through
    hidden_trans_video_multimodal (modeling.py:348-349)."""
    return ValorSpec(video_encoder="clip", txt_encoder="bert", vis_width=1024, vis_layers=24, patch=14, txt_width=768,
                     embed_dim=768, contra_dim=512)


def tiny_clip_bert_spec():
    """the code paths of clip_large_spec() (patch 14 -> a 588-wide conv GEMM, video width != hidden, Contra_head linears beside an
    untouched CLIP text tower) at unit-test size"""
    return ValorSpec(video_encoder="clip", txt_encoder="bert", vis_width=256, vis_layers=2, patch=14, resolution=56, txt_width=128,
                     txt_layers=1, clip_vocab=1200, embed_dim=128, contra_dim=64, aud_width=128, aud_layers=2, aud_inter=256,
                     melbins=32, target_len=64, aud_patch=16, hidden=128, layers=2, inter=256, vocab=1200, max_pos=64)


def large_spec():
    """BASELINE configs[3]: "VALOR-large (VideoSwin-L + BERT-large)". Not a shipped reference config (load_videoswin_model /
    load_bert_model accept base models only, modeling.py:578-587,618-625): the reference CLASSES with large hyper-parameters
    (SURVEY 8d config 4): Swin-L embed 192, heads 6/12/24/48, out 1536; BERT-large 1024 / 24 layers / 16 heads / 4096; AST stays
    768, so both hidden_trans_{video,audio}_multimodal exist (modeling.py:348-351)."""
    return ValorSpec(video_encoder="swin", txt_encoder="bert", swin_embed=192, swin_heads=(6, 12, 24, 48), hidden=1024, layers=24,
                     inter=4096)


def tiny_large_spec():
    """the WIDTHS of large_spec() (LayerNorm rows of 3072 in the last PatchMerging, hidden 1024 != audio 768 != video 1536) on a
    shallow stack: parity tests of the large configuration's code paths that finish in seconds"""
    return ValorSpec(video_encoder="swin", txt_encoder="bert", resolution=224, swin_embed=192, swin_depths=(1, 1, 2, 1),
                     swin_heads=(6, 12, 24, 48), hidden=1024, layers=2, inter=4096, aud_layers=1, vocab=4000, max_pos=64)


def swin_relative_position_index(window):
    """videoswin.py:112-126: index into the (2wd-1)(2wh-1)(2ww-1) bias table for every token pair of a FULL window;
    with lin(t) = d*(2wh-1)(2ww-1) + h*(2ww-1) + w it is lin(i) - lin(j) + lin(last token)."""
    wd, wh, ww = window
    d, h, w = torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")
    lin = (d * (2 * wh - 1) * (2 * ww - 1) + h * (2 * ww - 1) + w).reshape(-1)
    return lin[:, None] - lin[None, :] + int(lin[-1])


def tiny_spec():
    """Small architecture for fast unit tests (same code paths, every dim a multiple of 64)."""
    return ValorSpec(vis_width=128, vis_layers=2, patch=16, resolution=64, txt_width=128, txt_layers=2, ctx_len=77,
                     clip_vocab=1200, embed_dim=128, aud_width=128, aud_layers=2, aud_inter=256, melbins=32,
                     target_len=64, aud_patch=16, hidden=128, layers=2, inter=256, vocab=1200, max_pos=64)


def shallow_base_spec(video_encoder="clip"):
    """VALOR-base WIDTHS (768 / 512 / 12 heads, 224 px, 10 s audio geometry) on a 2-layer stack with a small vocabulary: a small
    invocation whose bf16 rounding behaves like the real model's (the tiny specs' 128-wide dot products average 6x fewer
    terms, so their bf16 noise is ~2.5x larger than anything the benchmarked configuration sees)."""
    if video_encoder == "swin":
        return ValorSpec(video_encoder="swin", txt_encoder="bert", swin_depths=(1, 1, 2, 1), aud_layers=2, layers=2, vocab=4000, max_pos=64)
    return ValorSpec(vis_layers=2, txt_layers=2, aud_layers=2, layers=2, vocab=4000, clip_vocab=4000, max_pos=64)


def _audio_bert_heads(spec, add):
    """AST + multimodal BERT + prediction head keys (both variants)."""
    H, AW = spec.hidden, spec.aud_width
    add("audio_embeddings.cls_token", (1, 1, AW))
    add("audio_embeddings.first_conv.weight", (AW, 1, spec.aud_patch, spec.aud_patch)); add("audio_embeddings.first_conv.bias", (AW,), "b")
    add("audio_embeddings.position_embeddings.weight", (spec.aud_tokens, AW))
    for i in range(spec.aud_layers):
        p = f"audio_encoder.layer.{i}."
        for j in range(4):
            add(p + f"attention.linears.{j}.weight", (AW, AW)); add(p + f"attention.linears.{j}.bias", (AW,), "b")
        add(p + "ff_layer.linear1.weight", (spec.aud_inter, AW)); add(p + "ff_layer.linear1.bias", (spec.aud_inter,), "b")
        add(p + "ff_layer.linear2.weight", (AW, spec.aud_inter)); add(p + "ff_layer.linear2.bias", (AW,), "b")
        add(p + "layernorm1.weight", (AW,), "g"); add(p + "layernorm1.bias", (AW,), "b")
        add(p + "layernorm2.weight", (AW,), "g"); add(p + "layernorm2.bias", (AW,), "b")
    add("audio_encoder.last_layernorm.weight", (AW,), "g"); add("audio_encoder.last_layernorm.bias", (AW,), "b")

    e = "multimodal_encoder.embeddings."
    add(e + "word_embeddings.weight", (spec.vocab, H)); add(e + "position_embeddings.weight", (spec.max_pos, H))
    add(e + "token_type_embeddings.weight", (2, H)); add(e + "prompt_embedding.weight", (1, H))
    add(e + "LayerNorm.weight", (H,), "g"); add(e + "LayerNorm.bias", (H,), "b")
    for i in range(spec.layers):
        p = f"multimodal_encoder.encoder.layer.{i}."
        for blk in ("attention.self",) + tuple(c + ".cross" for c in spec.cross_blocks):
            for n in ("query", "key", "value"):
                add(p + f"{blk}.{n}.weight", (H, H)); add(p + f"{blk}.{n}.bias", (H,), "b")
            out = blk.split(".")[0] + ".output."
            add(p + out + "dense.weight", (H, H)); add(p + out + "dense.bias", (H,), "b")
            add(p + out + "LayerNorm.weight", (H,), "g"); add(p + out + "LayerNorm.bias", (H,), "b")
        add(p + "intermediate.dense.weight", (spec.inter, H)); add(p + "intermediate.dense.bias", (spec.inter,), "b")
        add(p + "output.dense.weight", (H, spec.inter)); add(p + "output.dense.bias", (H,), "b")
        add(p + "output.LayerNorm.weight", (H,), "g"); add(p + "output.LayerNorm.bias", (H,), "b")
    add("multimodal_encoder.pooler.dense.weight", (H, H)); add("multimodal_encoder.pooler.dense.bias", (H,), "b")
    add("cls.dense.weight", (H, H)); add("cls.dense.bias", (H,), "b")
    add("cls.layernorm.weight", (H,), "g"); add("cls.layernorm.bias", (H,), "b")
    add("cls.decoder.weight", (spec.vocab, H), "tied"); add("cls.decoder.bias", (spec.vocab,), "b")


def state_dict_layout(spec: ValorSpec):
    """Ordered (key, shape, kind) of the VALOR state dict (CLIP or VideoSwin variant). kind: w (weight), b (bias), g (LN gain),
    s (scalar), tied / alias (same storage as another key), relidx (integer buffer)."""
    H, W, TW, AW, E = spec.hidden, spec.vis_width, spec.txt_width, spec.aud_width, spec.embed_dim
    L = []
    add = lambda k, s, kind="w": L.append((k, tuple(s), kind))
    add("video_type_embeddings", (1, 1, H)); add("audio_type_embeddings", (1, 1, H))
    add("video_frame_embedding", (1, 32, H)); add("audio_frame_embedding", (1, 32, H))
    add("contra_temp", (), "s")
    E, C = spec.embed_dim, spec.cdim
    if spec.video_encoder == "swin":
        assert spec.txt_encoder == "bert", "the reference loads CLIP as a whole: swin video + clip text is not a shipped combination"
        C0 = spec.swin_embed
        add("video_encoder.patch_embed.proj.weight", (C0, 3, 2, 4, 4)); add("video_encoder.patch_embed.proj.bias", (C0,), "b")
        add("video_encoder.patch_embed.norm.weight", (C0,), "g"); add("video_encoder.patch_embed.norm.bias", (C0,), "b")
        for li, (depth, nh) in enumerate(zip(spec.swin_depths, spec.swin_heads)):
            Cw = C0 * 2 ** li
            for bi in range(depth):
                p = f"video_encoder.layers.{li}.blocks.{bi}."
                add(p + "norm1.weight", (Cw,), "g"); add(p + "norm1.bias", (Cw,), "b")
                add(p + "attn.relative_position_bias_table", (spec.swin_table, nh))
                add(p + "attn.relative_position_index", (0,), "relidx")
                add(p + "attn.qkv.weight", (3 * Cw, Cw)); add(p + "attn.qkv.bias", (3 * Cw,), "b")
                add(p + "attn.proj.weight", (Cw, Cw)); add(p + "attn.proj.bias", (Cw,), "b")
                add(p + "norm2.weight", (Cw,), "g"); add(p + "norm2.bias", (Cw,), "b")
                add(p + "mlp.fc1.weight", (4 * Cw, Cw)); add(p + "mlp.fc1.bias", (4 * Cw,), "b")
                add(p + "mlp.fc2.weight", (Cw, 4 * Cw)); add(p + "mlp.fc2.bias", (Cw,), "b")
            if li + 1 < len(spec.swin_depths):
                p = f"video_encoder.layers.{li}.downsample."
                add(p + "reduction.weight", (2 * Cw, 4 * Cw)); add(p + "norm.weight", (4 * Cw,), "g"); add(p + "norm.bias", (4 * Cw,), "b")
        add("video_encoder.norm.weight", (spec.swin_out,), "g"); add("video_encoder.norm.bias", (spec.swin_out,), "b")
    else:
        # the whole CLIP model is one module (modeling.py:560-573): both towers are in the state dict whichever of them forward_pt uses
        add("clip_model.positional_embedding", (spec.ctx_len, TW)); add("clip_model.text_projection", (TW, E))
        add("clip_model.logit_scale", (), "s")
        add("clip_model.visual.class_embedding", (W,)); add("clip_model.visual.positional_embedding", (spec.vis_tokens, W))
        add("clip_model.visual.proj", (W, E)); add("clip_model.visual.conv1.weight", (W, 3, spec.patch, spec.patch))
        add("clip_model.visual.ln_pre.weight", (W,), "g"); add("clip_model.visual.ln_pre.bias", (W,), "b")

        def clip_blocks(prefix, width, n):
            for i in range(n):
                p = f"{prefix}.resblocks.{i}."
                add(p + "attn.in_proj_weight", (3 * width, width)); add(p + "attn.in_proj_bias", (3 * width,), "b")
                add(p + "attn.out_proj.weight", (width, width)); add(p + "attn.out_proj.bias", (width,), "b")
                add(p + "ln_1.weight", (width,), "g"); add(p + "ln_1.bias", (width,), "b")
                add(p + "mlp.c_fc.weight", (4 * width, width)); add(p + "mlp.c_fc.bias", (4 * width,), "b")
                add(p + "mlp.c_proj.weight", (width, 4 * width)); add(p + "mlp.c_proj.bias", (width,), "b")
                add(p + "ln_2.weight", (width,), "g"); add(p + "ln_2.bias", (width,), "b")
        clip_blocks("clip_model.visual.transformer", W, spec.vis_layers)
        add("clip_model.visual.ln_post.weight", (W,), "g"); add("clip_model.visual.ln_post.bias", (W,), "b")
        clip_blocks("clip_model.transformer", TW, spec.txt_layers)
        add("clip_model.token_embedding.weight", (spec.clip_vocab, TW))
        add("clip_model.ln_final.weight", (TW,), "g"); add("clip_model.ln_final.bias", (TW,), "b")
        add("clip_model.prompt_embedding.weight", (1, TW))

    _audio_bert_heads(spec, add)
    if spec.txt_encoder == "bert":
        # share_txt_and_multimodal (modeling.py:689-691): the text encoder IS the multimodal encoder, so the reference state
        # dict lists every multimodal_encoder.* tensor a second time under txt_encoder.* (same storage), right after cls.*
        for k, shape, kind in list(L):
            if k.startswith("multimodal_encoder."):
                L.append(("txt_encoder." + k[len("multimodal_encoder."):], shape, "alias"))
    if spec.video_dim != H:                                                           # modeling.py:348-349
        add("hidden_trans_video_multimodal.0.weight", (H, spec.video_dim)); add("hidden_trans_video_multimodal.0.bias", (H,), "b")
        add("hidden_trans_video_multimodal.1.weight", (H,), "g"); add("hidden_trans_video_multimodal.1.bias", (H,), "b")
    if AW != H:                                                                       # modeling.py:350-351
        add("hidden_trans_audio_multimodal.0.weight", (H, AW)); add("hidden_trans_audio_multimodal.0.bias", (H,), "b")
        add("hidden_trans_audio_multimodal.1.weight", (H,), "g"); add("hidden_trans_audio_multimodal.1.bias", (H,), "b")
    if not spec.clip_heads:                                                           # Contra_head, pretrain.py:93-97 (no bias)
        add("contra_head_t.linear.weight", (C, spec.txt_dim)); add("contra_head_v.linear.weight", (C, spec.video_dim))
    add("contra_head_a.linear.weight", (C, AW))
    if spec.contra_type == "fine":                                                    # pretrain.py:103-116
        for m in ("text", "video", "audio"):
            add(f"{m}_fine_weight.0.weight", (C, C)); add(f"{m}_fine_weight.0.bias", (C,), "b")
            add(f"{m}_fine_weight.2.weight", (1, C)); add(f"{m}_fine_weight.2.bias", (1,), "b")
    elif not spec.late_fusion:                                                        # pretrain.py:100-101
        add("va_fusion.weight", (C, 2 * C)); add("va_fusion.bias", (C,), "b")
    return L


def _bf16_exact(t):
    """nearest bf16-representable fp32 value: the SAME tensor then loads bit-identically into the fp32 reference / oracle and
    into the bf16 native model ("identical synthetic tensors" for the bf16 parity runs)"""
    return t.to(torch.bfloat16).to(torch.float32)


def make_state_dict(spec: ValorSpec, seed: int = 50, w_std: float = 0.02, bf16_exact: bool = False):
    """Seeded random-init state dict (fp32, CPU). LN gains 1 + 0.1 N(0,1), biases 0.02 N(0,1),
    weights w_std N(0,1); logit_scale = ln(1/0.07), contra_temp = 0.07; cls.decoder.weight is tied.
    bf16_exact: every floating tensor is rounded to a bf16-representable fp32 value (aliases stay aliases)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    import math
    for k, shape, kind in state_dict_layout(spec):
        if kind == "tied":
            sd[k] = sd["multimodal_encoder.embeddings.word_embeddings.weight"]
        elif kind == "alias":
            sd[k] = sd["multimodal_encoder." + k[len("txt_encoder."):]]
        elif kind == "relidx":
            sd[k] = swin_relative_position_index(spec.swin_window)
        elif kind == "s":
            sd[k] = torch.tensor(math.log(1 / 0.07) if "logit_scale" in k else 0.07)
        elif kind == "g":
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "b":
            sd[k] = 0.02 * torch.randn(shape, generator=g)
        else:
            sd[k] = w_std * torch.randn(shape, generator=g)
        if bf16_exact and kind not in ("tied", "alias", "relidx"):
            sd[k] = _bf16_exact(sd[k])
    return sd


def make_batch(spec: ValorSpec, batch: int, frames: int = 8, audio_slices: int = 2, txt_len: int = 32, seed: int = 50,
               bf16_exact: bool = False, questions: bool = False, question_len: int = 12, answers_per_question=None):
    """Synthetic batch with the schema of data/data.py:423-428 (valor_collate), CPU tensors.
    bf16_exact: pixels / spectrograms are bf16-representable fp32 values (see make_state_dict)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    video = torch.randn((batch, frames, 3, spec.resolution, spec.resolution), generator=g)
    audio = torch.randn((batch, audio_slices, spec.melbins, spec.target_len), generator=g)
    if bf16_exact:
        video, audio = _bf16_exact(video), _bf16_exact(audio)
    bert = torch.zeros((batch, txt_len), dtype=torch.long)
    clip = torch.zeros((batch, txt_len), dtype=torch.long)
    lo = min(1000, spec.vocab // 2)
    for i in range(batch):
        n = int(torch.randint(5, txt_len - 1, (1,), generator=g))
        bert[i, 0] = 101; bert[i, 1:1 + n] = torch.randint(lo, spec.vocab, (n,), generator=g); bert[i, 1 + n] = 102
        clip[i, 0] = spec.clip_vocab - 2
        clip[i, 1:1 + n] = torch.randint(min(1000, spec.clip_vocab // 2), spec.clip_vocab - 2, (n,), generator=g)
        clip[i, 1 + n] = spec.clip_vocab - 1
    out = {"ids": list(range(batch)), "txt_tokens": {"bert_tokens": bert, "clip_tokens": clip},
           "video_pixels": video, "audio_spectrograms": audio}
    if questions:       # video-QA schema (data/vqa.py:143-190): a question per clip, txt_tokens are then the answers, one answer per question
        qb = torch.zeros((batch, question_len), dtype=torch.long)
        for i in range(batch):
            n = int(torch.randint(3, question_len - 1, (1,), generator=g))
            qb[i, 0] = 101; qb[i, 1:1 + n] = torch.randint(lo, spec.vocab, (n,), generator=g); qb[i, 1 + n] = 102
        out.update(question_tokens={"bert_tokens": qb, "clip_tokens": torch.zeros_like(qb)}, answer_weights=[1.0] * batch,
                   answer_nums=[1] * batch, sample_num=[1] * batch)
        if answers_per_question is not None:      # image-QA schema (data/vqa.py:181-189): sum(nums) answer rows, a weight per row
            nums = [int(n) for n in answers_per_question]
            assert len(nums) == batch
            R = sum(nums)
            ab = torch.zeros((R, txt_len), dtype=torch.long)
            for r in range(R):
                n = int(torch.randint(2, txt_len - 1, (1,), generator=g))
                ab[r, 0] = 101; ab[r, 1:1 + n] = torch.randint(lo, spec.vocab, (n,), generator=g); ab[r, 1 + n] = 102
            w = torch.rand((R,), generator=g) * 0.7 + 0.3
            out.update(txt_tokens={"bert_tokens": ab, "clip_tokens": torch.zeros_like(ab)}, answer_weights=w, answer_nums=nums)
    return out


PROMPT_WORDS = ("describe the video with natural language predict masked tokens visual and audio cues project "
                "in common space answer question").split()


def synthetic_vocab(size: int = 30522):
    """WordPiece vocab with [PAD]=0 [UNK]=100 [CLS]=101 [SEP]=102 [MASK]=103 and the task-prompt words
    (model/pretrain.py:438,492,505,516) as whole tokens at 2000+ (or 200+ for tiny vocabularies)."""
    toks = [f"[unused{i}]" for i in range(size)]
    toks[0] = "[PAD]"; toks[100] = "[UNK]"; toks[101] = "[CLS]"; toks[102] = "[SEP]"; toks[103] = "[MASK]"
    base = 2000 if size > 4000 else 200
    for i, w in enumerate(dict.fromkeys(PROMPT_WORDS)):
        toks[base + i] = w
    return toks
