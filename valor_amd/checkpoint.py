"""Checkpoint interop of the reference's loading paths (host side, load time only; torch CPU ops are fine here):

  * adapt_pretrained_checkpoint: what train_utils.py::load_from_pretrained_dir does to a `model_step_N.pt` before
    VALOR.from_pretrained sees it (train_utils.py:146-168): frame embeddings beyond the pretraining sample count repeat the last
    trained frame, CLIP's visual positional embedding is bilinearly resized when the video resolution changes; the model options
    the pretraining run fixed are copied over the current ones (:134-144).
  * resize_clip_positional_embedding: the same resize for a bare `--checkpoint` (train.py:28-44).
The optimizer side (`optimizer_step_N.pt`, --resume) is FusedAdamW.load_reference_state_dict / reference_state_dict.
"""
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
