"""Checkpoint interop of the reference's loading paths (host side, load time only; torch CPU ops are fine here):

  * adapt_pretrained_checkpoint: what train_utils.py::load_from_pretrained_dir does to a `model_step_N.pt` before
    VALOR.from_pretrained sees it (train_utils.py:146-168): frame embeddings beyond the pretraining sample count repeat the last
    trained frame, CLIP's visual positional embedding is bilinearly resized when the video resolution changes; the model options
    the pretraining run fixed are copied over the current ones (:134-144).
  * resize_clip_positional_embedding: the same resize for a bare `--checkpoint` (train.py:28-44).
  * the component checkpoints the reference's constructor reads from ./pretrained_weights (the first pretraining run starts from
    them, train.py:17,55 with checkpoint = {}): AST (modeling.py:512-554, incl. the bilinear resize of its positional embedding
    :520-528), CLIP (:560-573 + clip.py:470-515), VideoSwin (:591-600) and BERT + prediction head (:613-660) key mappings into
    VALOR state-dict keys -- load_pretrained_components() merges them into ONE dict for VALOR.from_pretrained(opts, dict).
The optimizer side (`optimizer_step_N.pt`, --resume) is FusedAdamW.load_reference_state_dict / reference_state_dict.
"""
import os

import torch
import torch.nn.functional as F

# train_utils.py:134-141
COVER_CFG = ["audio_melbins", "audio_patch_size", "audio_mean", "audio_std", "audio_frame_shift", "audio_target_length",
             "video_encoder_type", "txt_encoder_type", "multimodal_encoder_type", "audio_encoder_type", "caption_type",
             "share_txt_and_multimodal", "contra_type", "multimodal_use_cross_attn", "fineweight_type", "has_vafusion_encoder",
             "late_fusion", "cross_attn_type", "task_pormpt_as_text", "use_task_prompt"]


def resize_clip_positional_embedding(checkpoint, video_resolution):
    """train_utils.py:153-168 / train.py:28-44: keep the cls row, resize the grid rows with F.interpolate(mode='bilinear')."""
    key = "clip_model.visual.positional_embedding"
    if key not in checkpoint:
        return checkpoint
    src = checkpoint[key]
    width = checkpoint["clip_model.visual.conv1.weight"].shape[0]
    patch = checkpoint["clip_model.visual.conv1.weight"].shape[-1]
    grid = round((src.shape[0] - 1) ** 0.5)
    new_grid = video_resolution // patch
    oth = src[1:].reshape(grid, grid, width).permute(2, 0, 1).unsqueeze(0)
    oth = F.interpolate(oth, (new_grid, new_grid), mode="bilinear")
    oth = oth[0].permute(1, 2, 0).reshape(-1, src.shape[-1])
    checkpoint[key] = torch.cat((src[0:1], oth), dim=0)
    return checkpoint


def adapt_pretrained_checkpoint(checkpoint, pretrain_cfg, opts):
    """checkpoint: reference-keyed state dict (a `module.` prefix is stripped); pretrain_cfg: the run's log/hps.json as a dict;
    opts: the current options (namespace or dict), updated in place like train_utils.py:142-144. Returns the adapted dict."""
    checkpoint = {k.replace("module.", ""): v for k, v in checkpoint.items()}
    get = (lambda k, d=None: opts.get(k, d)) if isinstance(opts, dict) else (lambda k, d=None: getattr(opts, k, d))
    for k in COVER_CFG:
        if k in pretrain_cfg:
            if isinstance(opts, dict):
                opts[k] = pretrain_cfg[k]
            else:
                setattr(opts, k, pretrain_cfg[k])
    if "video_frame_embedding" in checkpoint:
        n = pretrain_cfg["video_sample_num"]
        checkpoint["video_frame_embedding"][:, n:] = checkpoint["video_frame_embedding"][:, n - 1].clone()
    if "audio_frame_embedding" in checkpoint:
        n = pretrain_cfg["audio_sample_num"]
        checkpoint["audio_frame_embedding"][:, n:] = checkpoint["audio_frame_embedding"][:, n - 1].clone()
    if get("video_resolution") != pretrain_cfg["video_resolution"] and str(get("video_encoder_type", "")).startswith("clip"):
        resize_clip_positional_embedding(checkpoint, get("video_resolution"))
    return checkpoint


# ------------------------------------------------------------------------------------------ component checkpoints
def ast_to_valor(ast_weight, audio_melbins=64, audio_target_length=512, audio_patch_size=16):
    """initialize_audio_weights, modeling.py:512-554: the AudioSet AST checkpoint (timm DeiT keys under `module.v.`) -> VALOR keys.
    The packed qkv rows are split into linears.0/1/2, the distillation token is dropped, and the positional embedding (cls + 12 x 101
    patches of the 128-mel / 1024-frame pretraining geometry) is resized bilinearly to (melbins / patch) x (target_length / patch)."""
    W = ast_weight["module.v.cls_token"].shape[-1]
    out = {"audio_embeddings.cls_token": ast_weight["module.v.cls_token"],
           "audio_embeddings.first_conv.weight": ast_weight["module.v.patch_embed.proj.weight"],
           "audio_embeddings.first_conv.bias": ast_weight["module.v.patch_embed.proj.bias"]}
    pos = ast_weight["module.v.pos_embed"][0]
    oth = pos[2:].reshape(12, 101, -1).permute(2, 0, 1).unsqueeze(0)                  # [1, W, 12, 101]; row 1 is the distilled token
    th, tw = audio_melbins // audio_patch_size, audio_target_length // audio_patch_size
    oth = F.interpolate(oth, size=(th, tw), mode="bilinear").squeeze().permute(1, 2, 0).reshape(-1, W)
    out["audio_embeddings.position_embeddings.weight"] = torch.cat((pos[0:1], oth), dim=0)
    nl = len({k.split(".")[3] for k in ast_weight if k.startswith("module.v.blocks.")})
    for i in range(nl):
        s, d = f"module.v.blocks.{i}.", f"audio_encoder.layer.{i}."
        for j in range(3):
            out[d + f"attention.linears.{j}.weight"] = ast_weight[s + "attn.qkv.weight"][j * W:(j + 1) * W, :]
            out[d + f"attention.linears.{j}.bias"] = ast_weight[s + "attn.qkv.bias"][j * W:(j + 1) * W]
        out[d + "attention.linears.3.weight"] = ast_weight[s + "attn.proj.weight"]
        out[d + "attention.linears.3.bias"] = ast_weight[s + "attn.proj.bias"]
        for a, b in (("ff_layer.linear1", "mlp.fc1"), ("ff_layer.linear2", "mlp.fc2"), ("layernorm1", "norm1"), ("layernorm2", "norm2")):
            out[d + a + ".weight"] = ast_weight[s + b + ".weight"]
            out[d + a + ".bias"] = ast_weight[s + b + ".bias"]
    out["audio_encoder.last_layernorm.weight"] = ast_weight["module.v.norm.weight"]
    out["audio_encoder.last_layernorm.bias"] = ast_weight["module.v.norm.bias"]
    return out


def clip_to_valor(clip_sd, video_resolution):
    """load_clip_model + build_model, modeling.py:560-573 / clip.py:470-515: OpenAI CLIP state dict -> `clip_model.` keys in fp32, the
    visual positional embedding resized to video_resolution when it differs from the checkpoint's native grid (clip.py:481-491)."""
    # build_model resizes FIRST (on the checkpoint's own dtype), then loads into a model whose Conv / Linear / MultiheadAttention
    # tensors and projections were converted to fp16 (clip.py:446-467 convert_weights, :516-520), then .float() (modeling.py:573):
    # those tensors are fp16-representable in the reference (a no-op for the released fp16 checkpoints)
    out = {"clip_model." + k: v for k, v in clip_sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    out = resize_clip_positional_embedding(out, video_resolution)
    half = lambda k: (k.endswith(("conv1.weight", "in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight",
                                  "c_fc.bias", "c_proj.weight", "c_proj.bias")) or k in ("clip_model.text_projection", "clip_model.visual.proj"))
    return {k: (v.half().float() if half(k) else v.float()) for k, v in out.items()}


def swin_to_valor(swin_sd):
    """load_videoswin_model, modeling.py:591-600: SwinTransformer3D state dict -> `video_encoder.` keys (the integer
    relative_position_index buffers ride along; VALOR.load_state_dict regenerates and ignores them)."""
    return {"video_encoder." + k: v for k, v in swin_sd.items()}


def bert_to_valor(bert_weight, share_txt_and_multimodal=True):
    """load_bert_model, modeling.py:613-660: HF bert-base-uncased.bin -> `multimodal_encoder.` keys (`bert.` prefix stripped, the old
    gamma / beta LayerNorm names renamed, strict=False: the cross-attention blocks and the prompt embedding keep their
    initialisation) + the prediction head (`cls.predictions.*` -> cls.dense / cls.layernorm / cls.decoder.bias; the decoder
    weight is tied to the word embeddings, modeling.py:241)."""
    ren = {k.replace("bert.", "").replace("gamma", "weight").replace("beta", "bias"): v for k, v in bert_weight.items()}
    out = {"multimodal_encoder." + k: v for k, v in ren.items() if not k.startswith("cls.")}
    head = {"cls.dense.weight": "cls.predictions.transform.dense.weight", "cls.dense.bias": "cls.predictions.transform.dense.bias",
            "cls.layernorm.weight": "cls.predictions.transform.LayerNorm.weight", "cls.layernorm.bias": "cls.predictions.transform.LayerNorm.bias",
            "cls.decoder.bias": "cls.predictions.bias"}
    for dst, src in head.items():
        out[dst] = ren[src]
    # cls.decoder.weight: the reference loads it INTO the tied word-embedding tensor (cls.load_state_dict, modeling.py:650-653)
    out["multimodal_encoder.embeddings.word_embeddings.weight"] = ren["cls.predictions.decoder.weight"]
    return out


PRETRAINED_FILES = {   # modeling.py:514, 563-570, 592-599, 622
    "ast": "audioset_10_10_0.4593.pth", "bert_base_uncased": "bert-base-uncased.bin",
    "clip_vit_base_16": "clip-vit-base-16.pt", "clip_vit_base_32": "clip-vit-base-32.pt", "clip_vit_large_14": "clip-vit-large-14.pt",
    "clip_vit_large_14_336px": "clip-vit-large-14-336px.pt", "videoswin_small_k400_1k": "ckpt_video-swin.pt",
    "videoswin_base_k400_1k": "videoswin_base_k400_1k.pth", "videoswin_base_k400_22k": "videoswin_base_k400_22k.pth",
    "videoswin_base_k600_22k": "videoswin_base_k600_22k.pth",
}


def load_pretrained_components(opts, root="./pretrained_weights", load=None, jit_load=None):
    """What the reference's constructor pulls from ./pretrained_weights for `opts` (modeling.py:296-330), as ONE VALOR-keyed state dict
    for VALOR.from_pretrained(opts, sd): the component checkpoints named by video_encoder_type / txt_encoder_type /
    audio_encoder_type / multimodal_encoder_type. `load` / `jit_load` default to torch.load / torch.jit.load(...).state_dict()."""
    get = (lambda k, d=None: opts.get(k, d)) if isinstance(opts, dict) else (lambda k, d=None: getattr(opts, k, d))
    load = load or (lambda p: torch.load(p, map_location="cpu"))
    jit_load = jit_load or (lambda p: torch.jit.load(p, map_location="cpu").state_dict())
    sd = {}
    clip_type = None
    for t in (get("txt_encoder_type", "clip_vit_base_16"), get("video_encoder_type", "clip_vit_base_16")):       # modeling.py:298-303
        if t.startswith("clip"):
            clip_type = t
    if clip_type is not None:
        sd.update(clip_to_valor(jit_load(os.path.join(root, PRETRAINED_FILES[clip_type])), int(get("video_resolution", 224))))
    vt = get("video_encoder_type", "clip_vit_base_16")
    if vt.startswith("videoswin"):
        sd.update(swin_to_valor(load(os.path.join(root, PRETRAINED_FILES[vt]))))
    if get("audio_encoder_type", "ast").startswith("ast"):
        sd.update(ast_to_valor(load(os.path.join(root, PRETRAINED_FILES["ast"])), int(get("audio_melbins", 64)),
                               int(get("audio_target_length", 512)), int(get("audio_patch_size", 16))))
    if get("initial_multimodal", True):
        sd.update(bert_to_valor(load(os.path.join(root, PRETRAINED_FILES[get("multimodal_encoder_type", "bert_base_uncased")]))))
    return sd
