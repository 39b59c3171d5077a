"""Parameter table of the native VALOR model and its mapping to the reference checkpoint layout.

Internal parameters keep the reference's state-dict names except that q/k/v
projections are stored PACKED so the fused-QKV GEMM consumes them in place:
  multimodal_encoder.encoder.layer.N.attention.self.{query,key,value}.{weight,bias}   (bert.py:233-235)
        -> ...attention.self.qkv.{weight [3H,H], bias [3H]}
  multimodal_encoder.encoder.layer.N.cross_attn.cross.{key,value}.{weight,bias}        (bert.py:304-305)
        -> ...cross_attn.cross.kv.{weight [2H,H], bias [2H]}
  audio_encoder.layer.N.attention.linears.{0,1,2}.{weight,bias}                        (transformer.py:109)
        -> ...attention.qkv.{weight,bias}
CLIP's in_proj_weight is packed already (clip.py:176), and so is VideoSwin's attn.qkv (videoswin.py:129).
The VideoSwin + BERT-text variant (scripts/pretrain.sh:3-8) has video_encoder.* / contra_head_{t,v} / hidden_trans_video_multimodal
instead of clip_model.*; its shared text encoder appears in the reference state dict a second time as txt_encoder.* (same storage).
cls.decoder.weight is tied to the word embeddings (modeling.py:241) and is not a separate parameter.
state_dict()/load_state_dict() of the model translate to/from the reference keys, so checkpoints
(utils/save.py:45-64 `model_step_N.pt`) stay drop-in.
"""
from ..synth import ValorSpec

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def optimizer_group(ref_name, new_params_name=()):
    """optim/misc.py:13-64: 10 param groups; index = 2*family + (1 if no-decay).
    families: 0 basic, 1 new, 2 clip visual, 3 clip text, 4 decoder. Substring rules are case
    sensitive (AST `layernorm*`, `cls.layernorm`, CLIP `ln_*` weights ARE decayed: preserved)."""
    nd = any(t in ref_name for t in NO_DECAY)
    if "clip" in ref_name and "visual" in ref_name:
        fam = 2
    elif "clip" in ref_name:
        fam = 3
    elif "multimodal_encoder.decoder" in ref_name:
        fam = 4
    elif any(t in ref_name for t in new_params_name):
        fam = 1
    else:
        fam = 0
    return 2 * fam + (1 if nd else 0)


def param_table(spec: ValorSpec):
    """Ordered list of (internal_name, shape, [reference keys]) in forward-execution order
    (encoders first, heads last): arena order == reverse of gradient-ready order."""
    H, W, TW, AW, E, C = spec.hidden, spec.vis_width, spec.txt_width, spec.aud_width, spec.embed_dim, spec.cdim
    T = []

    def add(name, shape, refs=None):
        T.append((name, tuple(shape), list(refs) if refs else [name]))

    def clip_blocks(prefix, width, n):
        for i in range(n):
            p = f"{prefix}.resblocks.{i}."
            add(p + "ln_1.weight", (width,)); add(p + "ln_1.bias", (width,))
            add(p + "attn.in_proj_weight", (3 * width, width)); add(p + "attn.in_proj_bias", (3 * width,))
            add(p + "attn.out_proj.weight", (width, width)); add(p + "attn.out_proj.bias", (width,))
            add(p + "ln_2.weight", (width,)); add(p + "ln_2.bias", (width,))
            add(p + "mlp.c_fc.weight", (4 * width, width)); add(p + "mlp.c_fc.bias", (4 * width,))
            add(p + "mlp.c_proj.weight", (width, 4 * width)); add(p + "mlp.c_proj.bias", (width,))

    swin = spec.video_encoder == "swin"
    if swin:
        # ---- VideoSwin (videoswin.py:378-458; keys of SwinTransformer3D.state_dict(); relative_position_index is a buffer
        # that the model regenerates, see VALOR.state_dict)
        C0 = spec.swin_embed
        add("video_encoder.patch_embed.proj.weight", (C0, 3, 2, 4, 4)); add("video_encoder.patch_embed.proj.bias", (C0,))
        add("video_encoder.patch_embed.norm.weight", (C0,)); add("video_encoder.patch_embed.norm.bias", (C0,))
        for li, (depth, nh) in enumerate(zip(spec.swin_depths, spec.swin_heads)):
            Cw = C0 * 2 ** li
            for bi in range(depth):
                p = f"video_encoder.layers.{li}.blocks.{bi}."
                add(p + "norm1.weight", (Cw,)); add(p + "norm1.bias", (Cw,))
                add(p + "attn.qkv.weight", (3 * Cw, Cw)); add(p + "attn.qkv.bias", (3 * Cw,))
                add(p + "attn.relative_position_bias_table", (spec.swin_table, nh))
                add(p + "attn.proj.weight", (Cw, Cw)); add(p + "attn.proj.bias", (Cw,))
                add(p + "norm2.weight", (Cw,)); add(p + "norm2.bias", (Cw,))
                add(p + "mlp.fc1.weight", (4 * Cw, Cw)); add(p + "mlp.fc1.bias", (4 * Cw,))
                add(p + "mlp.fc2.weight", (Cw, 4 * Cw)); add(p + "mlp.fc2.bias", (Cw,))
            if li + 1 < len(spec.swin_depths):
                p = f"video_encoder.layers.{li}.downsample."
                add(p + "norm.weight", (4 * Cw,)); add(p + "norm.bias", (4 * Cw,)); add(p + "reduction.weight", (2 * Cw, 4 * Cw))
        add("video_encoder.norm.weight", (spec.swin_out,)); add("video_encoder.norm.bias", (spec.swin_out,))
    else:
        # ---- CLIP visual
        add("clip_model.visual.conv1.weight", (W, 3, spec.patch, spec.patch))
        add("clip_model.visual.class_embedding", (W,)); add("clip_model.visual.positional_embedding", (spec.vis_tokens, W))
        add("clip_model.visual.ln_pre.weight", (W,)); add("clip_model.visual.ln_pre.bias", (W,))
        clip_blocks("clip_model.visual.transformer", W, spec.vis_layers)
        add("clip_model.visual.ln_post.weight", (W,)); add("clip_model.visual.ln_post.bias", (W,))
        if spec.clip_heads:
            add("clip_model.visual.proj", (W, E))
    # ---- AST
    add("audio_embeddings.first_conv.weight", (AW, 1, spec.aud_patch, spec.aud_patch)); add("audio_embeddings.first_conv.bias", (AW,))
    add("audio_embeddings.cls_token", (1, 1, AW)); add("audio_embeddings.position_embeddings.weight", (spec.aud_tokens, AW))
    for i in range(spec.aud_layers):
        p = f"audio_encoder.layer.{i}."
        add(p + "layernorm1.weight", (AW,)); add(p + "layernorm1.bias", (AW,))
        add(p + "attention.qkv.weight", (3 * AW, AW), [p + f"attention.linears.{j}.weight" for j in range(3)])
        add(p + "attention.qkv.bias", (3 * AW,), [p + f"attention.linears.{j}.bias" for j in range(3)])
        add(p + "attention.linears.3.weight", (AW, AW)); add(p + "attention.linears.3.bias", (AW,))
        add(p + "layernorm2.weight", (AW,)); add(p + "layernorm2.bias", (AW,))
        add(p + "ff_layer.linear1.weight", (spec.aud_inter, AW)); add(p + "ff_layer.linear1.bias", (spec.aud_inter,))
        add(p + "ff_layer.linear2.weight", (AW, spec.aud_inter)); add(p + "ff_layer.linear2.bias", (AW,))
    add("audio_encoder.last_layernorm.weight", (AW,)); add("audio_encoder.last_layernorm.bias", (AW,))
    def clip_text_tower():
        add("clip_model.token_embedding.weight", (spec.clip_vocab, TW)); add("clip_model.positional_embedding", (spec.ctx_len, TW))
        add("clip_model.prompt_embedding.weight", (1, TW))
        clip_blocks("clip_model.transformer", TW, spec.txt_layers)
        add("clip_model.ln_final.weight", (TW,)); add("clip_model.ln_final.bias", (TW,))
        add("clip_model.text_projection", (TW, E))

    if spec.clip_heads:
        # ---- CLIP text
        clip_text_tower()
    if not swin:
        add("clip_model.logit_scale", ())                      # the contrastive temperature whenever the video encoder is CLIP (modeling.py:420-426)
    # ---- contrastive heads
    if not spec.clip_heads:                                    # Contra_head, pretrain.py:33-38,93-97 (no bias)
        add("contra_head_t.linear.weight", (C, spec.txt_dim)); add("contra_head_v.linear.weight", (C, spec.video_dim))
    add("contra_head_a.linear.weight", (C, AW))
    if spec.contra_type == "fine":                             # pretrain.py:103-116
        for m in ("text", "video", "audio"):
            add(f"{m}_fine_weight.0.weight", (C, C)); add(f"{m}_fine_weight.0.bias", (C,))
            add(f"{m}_fine_weight.2.weight", (1, C)); add(f"{m}_fine_weight.2.bias", (1,))
    elif not spec.late_fusion:                                 # pretrain.py:100-101
        add("va_fusion.weight", (C, 2 * C)); add("va_fusion.bias", (C,))
    add("contra_temp", ())
    # ---- decoder inputs
    if spec.video_dim != H:                                    # modeling.py:348-349
        add("hidden_trans_video_multimodal.0.weight", (H, spec.video_dim)); add("hidden_trans_video_multimodal.0.bias", (H,))
        add("hidden_trans_video_multimodal.1.weight", (H,)); add("hidden_trans_video_multimodal.1.bias", (H,))
    if AW != H:                                                # modeling.py:350-351
        add("hidden_trans_audio_multimodal.0.weight", (H, AW)); add("hidden_trans_audio_multimodal.0.bias", (H,))
        add("hidden_trans_audio_multimodal.1.weight", (H,)); add("hidden_trans_audio_multimodal.1.bias", (H,))
    add("video_frame_embedding", (1, 32, H)); add("video_type_embeddings", (1, 1, H))
    add("audio_frame_embedding", (1, 32, H)); add("audio_type_embeddings", (1, 1, H))
    # ---- BERT multimodal decoder
    e = "multimodal_encoder.embeddings."
    add(e + "word_embeddings.weight", (spec.vocab, H), [e + "word_embeddings.weight", "cls.decoder.weight"])
    add(e + "position_embeddings.weight", (spec.max_pos, H)); add(e + "token_type_embeddings.weight", (2, H))
    add(e + "prompt_embedding.weight", (1, H)); add(e + "LayerNorm.weight", (H,)); add(e + "LayerNorm.bias", (H,))
    for i in range(spec.layers):
        p = f"multimodal_encoder.encoder.layer.{i}."
        s = p + "attention.self."
        add(s + "qkv.weight", (3 * H, H), [s + f"{n}.weight" for n in ("query", "key", "value")])
        add(s + "qkv.bias", (3 * H,), [s + f"{n}.bias" for n in ("query", "key", "value")])
        o = p + "attention.output."
        add(o + "dense.weight", (H, H)); add(o + "dense.bias", (H,)); add(o + "LayerNorm.weight", (H,)); add(o + "LayerNorm.bias", (H,))
        for blk in spec.cross_blocks:          # "cross_attn", or "cross_attn_v" + "cross_attn_a" (bert.py:430-436)
            c = p + blk + ".cross."
            add(c + "query.weight", (H, H)); add(c + "query.bias", (H,))
            add(c + "kv.weight", (2 * H, H), [c + "key.weight", c + "value.weight"])
            add(c + "kv.bias", (2 * H,), [c + "key.bias", c + "value.bias"])
            o = p + blk + ".output."
            add(o + "dense.weight", (H, H)); add(o + "dense.bias", (H,)); add(o + "LayerNorm.weight", (H,)); add(o + "LayerNorm.bias", (H,))
        add(p + "intermediate.dense.weight", (spec.inter, H)); add(p + "intermediate.dense.bias", (spec.inter,))
        add(p + "output.dense.weight", (H, spec.inter)); add(p + "output.dense.bias", (H,))
        add(p + "output.LayerNorm.weight", (H,)); add(p + "output.LayerNorm.bias", (H,))
    add("multimodal_encoder.pooler.dense.weight", (H, H)); add("multimodal_encoder.pooler.dense.bias", (H,))
    add("cls.dense.weight", (H, H)); add("cls.dense.bias", (H,)); add("cls.layernorm.weight", (H,)); add("cls.layernorm.bias", (H,))
    add("cls.decoder.bias", (spec.vocab,))
    if not swin and not spec.clip_heads:
        # CLIP video encoder beside a BERT text encoder (config/pretrain-VALOR-large.json): the reference constructs the WHOLE CLIP
        # model (modeling.py:560-573), so its text tower and both CLIP projections are checkpoint tensors and optimizer-group members
        # that forward_pt never touches. They live at the very END of the arena: their chunks stay inactive in the fused optimizer
        # and their gradient buckets are never launched.
        add("clip_model.visual.proj", (W, E))
        clip_text_tower()
    return T
