"""TEST INFRASTRUCTURE (not product code): run the UNMODIFIED reference VALOR from /root/reference on CPU.

Only usable where /root/reference exists (this build container). It is used to
  * pin the restated oracle (oracle/valor_oracle.py) against the real reference, and
  * generate the golden vectors committed under tests/golden/ (oracle/make_goldens.py).
Nothing here is imported by valor_amd.

Recipe: stub the arithmetic-free imports (apex FusedLayerNorm -> nn.LayerNorm, ipdb,
boto3, ftfy, easydict, tensorboardX, torchvision, toolz), make Tensor.cuda a no-op, single-rank gloo
group for ddp_allgather*, a temp cwd holding ./pretrained_weights/{bert config, vocab}, and
torch.load / torch.jit.load patched to hand back synthetic "pretrained" dicts of the right shapes
(values irrelevant: every tensor is then overwritten from a seeded state dict, strict=True).
"""
import json
import os
import sys
import tempfile

import torch

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


_installed = False


def _install():
    global _installed
    if _installed:
        return
    sys.path.insert(0, os.path.join(HERE, "_stubs"))
    sys.path.insert(0, REF_ROOT)
    import logging
    logging.disable(logging.INFO)   # the reference logs every missing key at INFO
    torch.Tensor.cuda = lambda self, *a, **k: self
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))    # per process: several harness users may coexist
        dist.init_process_group("gloo", rank=0, world_size=1)
    _installed = True


def default_opts(**over):
    _install()
    from easydict import EasyDict
    o = dict(
        video_resolution=224, audio_melbins=64, audio_patch_size=16, audio_frame_shift=10, audio_target_length=512,
        audio_mean=-4.2677393, audio_std=4.5689974,
        video_encoder_type="clip_vit_base_16", txt_encoder_type="clip_vit_base_16", audio_encoder_type="ast",
        multimodal_encoder_type="bert_base_uncased", share_txt_and_multimodal=True, multimodal_use_cross_attn=True,
        contra_type="fine", caption_type="unimlm", cross_attn_type="va_concate", feature_pooling_type="none",
        initial_multimodal=True, initial_vision=True, frozen_vision=False, frozen_multimodal=False, checkpointing=False,
        init_clip_head=True, max_generation_len=30, beam_size=3, beam_size_qa=1, label_smoothing=0.0,
        evaluate_ret_text=False, scst_finetuning=False, full_masker=False, contra_loss_ratio=1.0, fineweight_type="one",
        use_task_prompt=False, late_fusion=False, dual_softmax=False, contra_dim=512,
    )
    o.update(over)
    return EasyDict(o)


def _fake_bert_sd(cfg):
    H, L, I, V = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"], cfg["vocab_size"]
    z = torch.zeros
    sd = {
        "bert.embeddings.word_embeddings.weight": z(V, H),
        "bert.embeddings.position_embeddings.weight": z(cfg["max_position_embeddings"], H),
        "bert.embeddings.token_type_embeddings.weight": z(cfg["type_vocab_size"], H),
        "bert.embeddings.LayerNorm.gamma": z(H), "bert.embeddings.LayerNorm.beta": z(H),
        "cls.predictions.transform.dense.weight": z(H, H), "cls.predictions.transform.dense.bias": z(H),
        "cls.predictions.transform.LayerNorm.weight": z(H), "cls.predictions.transform.LayerNorm.bias": z(H),
        "cls.predictions.decoder.weight": z(V, H), "cls.predictions.bias": z(V),
    }
    return sd


# CLIP checkpoints the reference's load_clip_model accepts (modeling.py:560-573); build_model derives every hyper-parameter from
# tensor SHAPES and layer COUNTS (clip.py:470-515), so `layers` can be cut for a fast pin: (vision width, vision layers, patch,
# native grid, joint embed dim, text width, text layers)
CLIP_KINDS = {
    "clip_vit_base_16": (768, 12, 16, 14, 512, 512, 12),
    "clip_vit_large_14_336px": (1024, 24, 14, 24, 768, 768, 12),       # 336 px native: the positional embedding is resized to
    "clip_vit_large_14": (1024, 24, 14, 16, 768, 768, 12),             # opts.video_resolution at construction (clip.py:481-491)
}


def _fake_clip_sd(kind="clip_vit_base_16", vision_layers=None, text_layers=None):
    # values are overwritten later from the seeded canonical state dict
    z = torch.zeros
    W, VL, P, G, E, TW, TL = CLIP_KINDS[kind]
    VL = vision_layers or VL
    TL = text_layers or TL
    sd = {"visual.conv1.weight": z(W, 3, P, P), "visual.class_embedding": z(W),
          "visual.positional_embedding": z(G * G + 1, W), "visual.proj": z(W, E),
          "visual.ln_pre.weight": z(W), "visual.ln_pre.bias": z(W), "visual.ln_post.weight": z(W),
          "visual.ln_post.bias": z(W), "text_projection": z(TW, E), "positional_embedding": z(77, TW),
          "token_embedding.weight": z(49408, TW), "ln_final.weight": z(TW), "ln_final.bias": z(TW),
          "logit_scale": z(())}
    for pre, w, n in (("visual.transformer", W, VL), ("transformer", TW, TL)):
        for i in range(n):
            p = f"{pre}.resblocks.{i}."
            sd[p + "attn.in_proj_weight"] = z(3 * w, w); sd[p + "attn.in_proj_bias"] = z(3 * w)
            sd[p + "attn.out_proj.weight"] = z(w, w); sd[p + "attn.out_proj.bias"] = z(w)
            sd[p + "ln_1.weight"] = z(w); sd[p + "ln_1.bias"] = z(w); sd[p + "ln_2.weight"] = z(w); sd[p + "ln_2.bias"] = z(w)
            sd[p + "mlp.c_fc.weight"] = z(4 * w, w); sd[p + "mlp.c_fc.bias"] = z(4 * w)
            sd[p + "mlp.c_proj.weight"] = z(w, 4 * w); sd[p + "mlp.c_proj.bias"] = z(w)
    return sd


def _fake_ast_sd():
    z = torch.zeros
    sd = {"module.v.cls_token": z(1, 1, 768), "module.v.dist_token": z(1, 1, 768),
          "module.v.patch_embed.proj.weight": z(768, 1, 16, 16), "module.v.patch_embed.proj.bias": z(768),
          "module.v.pos_embed": z(1, 2 + 12 * 101, 768), "module.v.norm.weight": z(768), "module.v.norm.bias": z(768)}
    for i in range(12):
        p = f"module.v.blocks.{i}."
        sd[p + "attn.qkv.weight"] = z(2304, 768); sd[p + "attn.qkv.bias"] = z(2304)
        sd[p + "attn.proj.weight"] = z(768, 768); sd[p + "attn.proj.bias"] = z(768)
        sd[p + "mlp.fc1.weight"] = z(3072, 768); sd[p + "mlp.fc1.bias"] = z(3072)
        sd[p + "mlp.fc2.weight"] = z(768, 3072); sd[p + "mlp.fc2.bias"] = z(768)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = z(768); sd[p + n + ".bias"] = z(768)
    return sd


BERT_CFG = dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1, hidden_size=768,
                initializer_range=0.02, intermediate_size=3072, max_position_embeddings=512, num_attention_heads=12,
                num_hidden_layers=12, type_vocab_size=2, vocab_size=30522)


def write_vocab(path):
    """30522-entry WordPiece vocab with [CLS]=101 [SEP]=102 [MASK]=103 (asserted at data/data.py:59-60)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from valor_amd.synth import synthetic_vocab
    with open(path, "w") as f:
        f.write("\n".join(synthetic_vocab(30522)) + "\n")


class _FakeJit:
    def __init__(self, sd):
        self._sd = sd

    def state_dict(self):
        return self._sd


def build_reference(opts=None, state_dict=None, dropout=0.0, clip_layers=None, bert_layers=None, fakes=None):
    """Instantiate the reference VALOR on CPU (fp32). state_dict: canonical VALOR state dict to load (strict).
    clip_layers = (vision, text) / bert_layers: build shallower component stacks (the reference derives the CLIP depth from the
    checkpoint's keys and the BERT depth from its json config), for pins that have to finish in seconds.
    fakes: optional {"bert" | "ast" | "clip" | "swin": state dict} handed to the reference's torch.load / torch.jit.load calls in
    place of the zero-filled stand-ins (to pin the component-checkpoint key mappings of valor_amd.checkpoint)."""
    fakes = fakes or {}
    _install()
    opts = opts or default_opts()
    bert_cfg = dict(BERT_CFG, num_hidden_layers=bert_layers) if bert_layers else BERT_CFG
    clip_kind = next((t for t in (opts.txt_encoder_type, opts.video_encoder_type) if t.startswith("clip")), "clip_vit_base_16")
    tmp = tempfile.mkdtemp(prefix="valor_ref_")
    os.makedirs(os.path.join(tmp, "pretrained_weights"))
    with open(os.path.join(tmp, "pretrained_weights", "bert_base_uncased_config.json"), "w") as f:
        json.dump(bert_cfg, f)
    write_vocab(os.path.join(tmp, "pretrained_weights", "bert-base-uncased-vocab.txt"))
    cwd = os.getcwd()
    os.chdir(tmp)
    real_load, real_jit = torch.load, torch.jit.load

    def fake_load(path, *a, **k):
        p = str(path)
        if "bert-base-uncased.bin" in p:
            return fakes["bert"] if "bert" in fakes else _fake_bert_sd(bert_cfg)
        if "audioset" in p:
            return fakes["ast"] if "ast" in fakes else _fake_ast_sd()
        if ("videoswin" in p or "video-swin" in p) and "swin" in fakes:
            return fakes["swin"]
        if "videoswin" in p or "video-swin" in p:
            from model.videoswin import SwinTransformer3D
            if "small" in opts.video_encoder_type:
                return SwinTransformer3D(embed_dim=96, num_heads=[3, 6, 12, 24]).state_dict()
            return SwinTransformer3D(embed_dim=128, num_heads=[4, 8, 16, 32]).state_dict()
        return real_load(path, *a, **k)

    torch.load = fake_load
    torch.jit.load = lambda path, *a, **k: _FakeJit(fakes["clip"] if "clip" in fakes else _fake_clip_sd(clip_kind, *(clip_layers or (None, None))))
    try:
        from model.pretrain import VALOR
        model = VALOR.from_pretrained(opts, {})
    finally:
        torch.load, torch.jit.load = real_load, real_jit
        os.chdir(cwd)
    model = model.float()
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not missing and not unexpected, (missing[:5], unexpected[:5])
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = dropout
        if type(m).__name__ == "DropPath":          # videoswin.py:50-55 stochastic depth: off whenever dropout is off
            m.drop_prob = m.drop_prob if dropout > 0 else 0.0
    return model


if __name__ == "__main__":
    over = dict(a.split("=", 1) for a in sys.argv[1:])
    m = build_reference(default_opts(**{k: (v == "True" if v in ("True", "False") else v) for k, v in over.items()}) if over else None)
    sd = m.state_dict()
    n = sum(p.numel() for p in m.parameters())
    print("params", n, "tensors", len(sd))
    with open("/tmp/ref_keys.txt", "w") as f:
        for k, v in sd.items():
            f.write(f"{k} {tuple(v.shape)} {v.dtype}\n")
    print("named_parameters", len(list(m.named_parameters())))
