"""The reference's option handling for the pretraining path: argparse defaults (train_utils.py:599-699) overlaid by a
`--config` JSON file, command-line values winning over the file (utils/misc.py:26-36 parse_with_config). The result is the
`opts` object VALOR.from_pretrained / FusedAdamW / TrainEngine read, so config/pretrain-VALOR-{base,large}.json drive this
repo unchanged.

Both shipped JSON files lack their final closing brace (json.load fails on them as shipped): load_config closes what is
left open at the end of the text instead of rejecting the file.
"""
import json
from types import SimpleNamespace

# argparse defaults of the options this path reads (train_utils.py:599-699); everything else in a config file is carried through
DEFAULTS = dict(
    video_resolution=224, audio_melbins=64, audio_patch_size=16, audio_frame_shift=10, audio_target_length=512,
    audio_mean=-4.2677393, audio_std=4.5689974, gradient_accumulation_steps=1, learning_rate=None, clip_lr=5e-7, clip_lr_text=5e-7,
    optim="adam", betas=[0.9, 0.98], dropout=0.1, weight_decay=0.01, decoder_lr=-1, grad_norm=5.0, warmup_ratio=0.1, seed=42,
    fp16=True, scheduler="warmup_linear", contra_type="fine", caption_type="unimlm", cross_attn_type="va_concate",
    max_generation_len=30, use_task_prompt=False, init_clip_head=True, late_fusion=False, checkpointing=False,
    frozen_multimodal=False, frozen_vision=False, video_encoder_type="clip_vit_base_16", txt_encoder_type="clip_vit_base_16",
    audio_encoder_type="ast", feature_pooling_type="none", multimodal_encoder_type="bert_base_uncased", num_train_steps=0,
    dual_softmax=False, contra_loss_ratio=1.0, dataset_mix_type="random", initial_multimodal=True, share_txt_and_multimodal=True,
    multimodal_use_cross_attn=True, new_lr=0.0, full_masker=False, new_params_name=[], beam_size=3, beam_size_qa=1, contra_dim=512,
    label_smoothing=0.0, fineweight_type="one", evaluate_ret_text=False, scst_finetuning=False,
)


def parse_json_lenient(text):
    """json.loads, closing braces / brackets the file leaves open at its end (the shipped pretrain-VALOR-*.json do)"""
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        pass
    stack, in_str, esc = [], False, False
    for ch in text:
        if in_str:
            if esc:
                esc = False
            elif ch == "\\":
                esc = True
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
        elif ch in "{[":
            stack.append("}" if ch == "{" else "]")
        elif ch in "}]":
            if not stack or stack.pop() != ch:
                raise ValueError("config file: unbalanced brackets")
    return json.loads(text.rstrip() + "".join(reversed(stack)))


def load_config(config=None, overrides=None):
    """opts = argparse defaults <- config file (path, JSON text or dict) <- explicit overrides (the command line)."""
    opts = dict(DEFAULTS)
    if config is not None:
        if isinstance(config, dict):
            cfg = config
        else:
            text = config if config.lstrip().startswith("{") else open(config).read()
            cfg = parse_json_lenient(text)
        opts.update(cfg)
    if overrides:
        opts.update(overrides)
    return SimpleNamespace(**opts)


def train_tasks(opts):
    """(task string, per-dataset batch size) of every training mix entry (data_cfg.train[*], data/loader.py builds one loader each)"""
    return [(d["task"], d.get("batch_size")) for d in getattr(opts, "data_cfg", {}).get("train", [])]
