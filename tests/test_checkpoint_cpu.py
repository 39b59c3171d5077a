"""CPU: checkpoint interop of the reference's loading paths, host side only.
  * valor_amd.checkpoint.adapt_pretrained_checkpoint against the UNMODIFIED body of train_utils.py::load_from_pretrained_dir
    (executed from the reference file where /root/reference exists; train_utils itself cannot be imported without the data stack);
  * FusedAdamW.load_reference_state_dict / reference_state_dict against the reference's own optimizer
    (optim/misc.py::build_optimizer + optim/adamw.py::AdamW, torch Optimizer.state_dict format of optimizer_step_N.pt)."""
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_harness  # noqa: E402
from valor_amd import synth  # noqa: E402
from valor_amd.checkpoint import adapt_pretrained_checkpoint, resize_clip_positional_embedding  # noqa: E402

needs_ref = pytest.mark.skipif(not ref_harness.available(), reason="/root/reference not present")


def _fake_checkpoint(seed=0, prefix=""):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(s, generator=g)
    return {prefix + "video_frame_embedding": r(1, 32, 24), prefix + "audio_frame_embedding": r(1, 32, 24),
            prefix + "clip_model.visual.conv1.weight": r(40, 3, 16, 16), prefix + "clip_model.visual.positional_embedding": r(14 * 14 + 1, 40),
            prefix + "cls.dense.weight": r(24, 24)}


HPS = {"video_sample_num": 3, "audio_sample_num": 1, "video_resolution": 224, "video_encoder_type": "clip_vit_base_16",
       "txt_encoder_type": "clip_vit_base_16", "contra_type": "fine", "use_task_prompt": True, "audio_melbins": 64, "learning_rate": 1.0}


def test_adapt_pretrained_checkpoint():
    ck = _fake_checkpoint(prefix="module.")
    opts = types.SimpleNamespace(video_resolution=448, video_encoder_type="x", use_task_prompt=False, learning_rate=3e-4)
    out = adapt_pretrained_checkpoint({k: v.clone() for k, v in ck.items()}, HPS, opts)
    assert all(not k.startswith("module.") for k in out)
    assert opts.video_encoder_type == "clip_vit_base_16" and opts.use_task_prompt is True and opts.learning_rate == 3e-4   # only COVER_CFG keys
    v = out["video_frame_embedding"]
    assert torch.equal(v[:, :3], ck["module.video_frame_embedding"][:, :3]) and all(torch.equal(v[:, i], v[:, 2]) for i in range(3, 32))
    a = out["audio_frame_embedding"]
    assert all(torch.equal(a[:, i], a[:, 0]) for i in range(1, 32))
    pe = out["clip_model.visual.positional_embedding"]
    assert pe.shape == (28 * 28 + 1, 40) and torch.equal(pe[0], ck["module.clip_model.visual.positional_embedding"][0])
    same = resize_clip_positional_embedding({k[7:]: v.clone() for k, v in ck.items()}, 224)["clip_model.visual.positional_embedding"]
    assert torch.allclose(same, ck["module.clip_model.visual.positional_embedding"], atol=1e-6)     # same grid: identity


@needs_ref
def test_adapt_pretrained_checkpoint_matches_reference_function(tmp_path):
    src = open(os.path.join(ref_harness.REF_ROOT, "train_utils.py")).read()
    a = src.index("def load_from_pretrained_dir(opts):")
    b = src.index("    return checkpoint", a) + len("    return checkpoint")
    ns = {"os": os, "torch": torch, "json": json, "F": torch.nn.functional, "edict": lambda d: types.SimpleNamespace(**d, __contains__=None),
          "LOGGER": types.SimpleNamespace(info=lambda *a, **k: None)}

    class Cfg(dict):                                   # EasyDict stand-in: attribute and item access, `in`
        __getattr__ = dict.__getitem__
    ns["edict"] = Cfg
    exec(compile(src[a:b], "train_utils.py[load_from_pretrained_dir]", "exec"), ns)
    ck = _fake_checkpoint(seed=3, prefix="module.")
    os.makedirs(tmp_path / "ckpt"); os.makedirs(tmp_path / "log")
    torch.save(ck, tmp_path / "ckpt" / "model_step_7.pt")
    torch.save(_fake_checkpoint(seed=9), tmp_path / "ckpt" / "model_step_3.pt")
    json.dump(HPS, open(tmp_path / "log" / "hps.json", "w"))
    mk = lambda: types.SimpleNamespace(pretrain_dir=str(tmp_path), pretrain_step=None, video_resolution=448, video_encoder_type="clip_vit_base_16",
                                       use_task_prompt=False, contra_type="coarse")
    o_ref, o_mine = mk(), mk()
    ref = ns["load_from_pretrained_dir"](o_ref)                       # picks the largest step, strips module., adapts
    mine = adapt_pretrained_checkpoint(torch.load(tmp_path / "ckpt" / "model_step_7.pt"), HPS, o_mine)
    assert set(ref) == set(mine)
    for k in ref:
        assert torch.equal(ref[k], mine[k]), k
    assert vars(o_ref) == vars(o_mine)


class _Named:
    """stand-in for the reference model: build_optimizer only walks named_parameters()"""
    def __init__(self, sd, spec):
        self.params = [(k, torch.nn.Parameter(sd[k].clone())) for k, _, kind in synth.state_dict_layout(spec) if kind not in ("alias", "relidx", "tied")]

    def named_parameters(self):
        return iter(self.params)


@needs_ref
@pytest.mark.parametrize("variant", ["clip", "swin"])
def test_optimizer_state_dict_is_the_references(variant):
    ref_harness._install()
    from easydict import EasyDict
    from optim.misc import build_optimizer
    from valor_amd.model.valor import VALOR
    from valor_amd.optim import FusedAdamW
    spec = synth.tiny_spec() if variant == "clip" else synth.tiny_swin_spec()
    sd = synth.make_state_dict(spec, seed=2)
    opts = EasyDict(learning_rate=1e-4, weight_decay=0.01, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=2e-4, decoder_lr=-1,
                    new_params_name=["contra_head", "fine_weight"], optim="adamw", betas=[0.9, 0.98])
    fake = _Named(sd, spec)
    ropt = build_optimizer(fake, opts)
    g = torch.Generator().manual_seed(5)
    # some tensors never get a gradient (task-dependent unused parameters); q / k / v of one projection always go together
    from valor_amd.model.params import param_table
    unused = {r for i, (_, _, refs) in enumerate(param_table(spec)) if i % 7 == 3 for r in refs}
    for step in range(2):
        for k, p in fake.params:
            p.grad = None if k in unused else 0.01 * torch.randn(p.shape, generator=g)
        ropt.step()
    ref_sd = ropt.state_dict()

    model = VALOR({"new_params_name": ["contra_head", "fine_weight"]}, spec=spec, dtype=torch.float32, device="cpu")
    mine = FusedAdamW(model, opts)
    groups = mine.reference_param_groups()
    assert [len(x) for x in groups] == [len(pg["params"]) for pg in ref_sd["param_groups"]]
    order = [r for names in groups for r in names]
    ref_order = {id(p): k for k, p in fake.params}
    assert order == [ref_order[id(p)] for pg in ropt.param_groups for p in pg["params"]]          # same parameter indexing
    mine.load_reference_state_dict(ref_sd)
    back = mine.reference_state_dict()
    assert set(back["state"]) == set(ref_sd["state"])
    for i, st in ref_sd["state"].items():
        assert back["state"][i]["step"] == st["step"]
        assert torch.equal(back["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(back["state"][i]["exp_avg_sq"], st["exp_avg_sq"]), order[i]
    for a, b in zip(back["param_groups"], ref_sd["param_groups"]):
        assert a["params"] == b["params"] and a["weight_decay"] == b["weight_decay"] and abs(a["lr"] - b["lr"]) < 1e-12
    # packed tensors: the fused q|k|v moments are the reference's three tensors stacked
    name = "multimodal_encoder.encoder.layer.0.attention.self.qkv.weight"
    o, n, shape = model.arena.offsets[name]
    H = spec.hidden
    ik = order.index("multimodal_encoder.encoder.layer.0.attention.self.key.weight")
    if ik in ref_sd["state"]:
        assert torch.equal(mine.exp_avg[o:o + n].view(shape)[H:2 * H], ref_sd["state"][ik]["exp_avg"])
    # and the reference's optimizer accepts what we export
    ropt2 = build_optimizer(_Named(sd, spec), opts)
    ropt2.load_state_dict(back)
    assert len(ropt2.state_dict()["state"]) == len(ref_sd["state"])


@needs_ref
@pytest.mark.parametrize("variant", ["clip", "swin", "clip_large_bert"])
def test_component_checkpoint_mappings_are_the_reference_constructors(variant):
    """The reference's constructor reads CLIP / VideoSwin / AST / BERT checkpoints from ./pretrained_weights and renames / splits /
    resizes them into its modules (modeling.py:512-554 incl. the AST positional-embedding interpolation :520-528, :560-573 +
    clip.py:470-515 incl. the 336 -> 224 px CLIP resize, :591-600, :613-660). Component files with seeded RANDOM values go through the
    unmodified constructor (ref_harness hands them to its torch.load / torch.jit.load calls) and through
    valor_amd.checkpoint.load_pretrained_components: every tensor the mapping produces must be bit-identical to the constructed
    reference model's, and every reference tensor the mapping does not produce must be one the reference leaves at its init."""
    from valor_amd.checkpoint import load_pretrained_components
    g = torch.Generator().manual_seed(17)
    rnd = lambda sd: {k: (torch.randn(v.shape, generator=g) if v.is_floating_point() else v) for k, v in sd.items()}
    bert_cfg = dict(ref_harness.BERT_CFG, num_hidden_layers=2)
    kind = {"clip": "clip_vit_base_16", "swin": "clip_vit_base_16", "clip_large_bert": "clip_vit_large_14_336px"}[variant]
    fakes = {"bert": rnd(ref_harness._fake_bert_sd(bert_cfg)), "ast": rnd(ref_harness._fake_ast_sd()),
             "clip": rnd(ref_harness._fake_clip_sd(kind, 2, 1))}
    # a real bert-base-uncased.bin: every embedding / encoder / pooler tensor under `bert.`, LayerNorms under the old gamma / beta names
    import dataclasses as _dc
    for k, shape, kind in synth.state_dict_layout(_dc.replace(synth.base_spec(), layers=2)):
        if k.startswith("multimodal_encoder.") and "cross_attn" not in k and "prompt_embedding" not in k:
            name = "bert." + k[len("multimodal_encoder."):].replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
            fakes["bert"][name] = torch.randn(shape, generator=g)
    over = {"clip": {}, "swin": dict(video_encoder_type="videoswin_base_k400_22k", txt_encoder_type="bert_base_uncased"),
            "clip_large_bert": dict(video_encoder_type="clip_vit_large_14_336px", txt_encoder_type="bert_base_uncased")}[variant]
    ropts = ref_harness.default_opts(**over)
    if variant == "swin":
        ref_harness._install()
        from model.videoswin import SwinTransformer3D
        fakes["swin"] = rnd(SwinTransformer3D(embed_dim=128, num_heads=[4, 8, 16, 32]).state_dict())
    ref = ref_harness.build_reference(ropts, state_dict=None, bert_layers=2, fakes={k: {a: b.clone() for a, b in v.items()} for k, v in fakes.items()})
    rsd = ref.state_dict()
    files = {"bert-base-uncased.bin": fakes["bert"], "audioset_10_10_0.4593.pth": fakes["ast"]}
    if "swin" in fakes:
        files["videoswin_base_k400_22k.pth"] = fakes["swin"]
    mine = load_pretrained_components(ropts, root="", load=lambda p: {k: v.clone() for k, v in files[os.path.basename(p)].items()},
                                      jit_load=lambda p: {k: v.clone() for k, v in fakes["clip"].items()})
    produced = 0
    for k, v in mine.items():
        if k not in rsd:          # what the reference's strict=False loads drop: the distillation token, HF head / pooler extras
            assert k.startswith("multimodal_encoder.cls.") or "distill" in k or "seq_relationship" in k, k
            continue
        assert rsd[k].shape == v.shape and torch.equal(rsd[k].float(), v.float()), k
        produced += 1
    assert produced > {"clip": 250, "swin": 550, "clip_large_bert": 250}[variant]       # swin: 351 VideoSwin + 197 AST + 45 two-layer-BERT tensors = 593
    if variant != "swin":
        assert mine["clip_model.visual.positional_embedding"].shape[0] == (224 // (16 if variant == "clip" else 14)) ** 2 + 1
    assert mine["audio_embeddings.position_embeddings.weight"].shape == (129, 768)
    # and the merged dict loads into the native model's layout (CPU construction: tables only)
    from valor_amd.model.valor import VALOR
    import dataclasses
    spec = {"clip": synth.base_spec(), "swin": synth.swin_spec(), "clip_large_bert": synth.clip_large_spec()}[variant]
    spec = dataclasses.replace(spec, layers=2, **({} if variant == "swin" else dict(vis_layers=2, txt_layers=1)))
    model = VALOR({k: v for k, v in vars(ropts).items()} if hasattr(ropts, "__dict__") else dict(ropts), spec=spec, dtype=torch.float32, device="cpu")
    missing, unexpected = model.load_state_dict(mine, strict=False)
    fresh = ("contra_head_t", "contra_head_v", "contra_head_a", "text_fine_weight", "video_fine_weight", "audio_fine_weight", "contra_temp",
             "video_type_embeddings", "audio_type_embeddings", "video_frame_embedding", "audio_frame_embedding", "hidden_trans_video_multimodal",
             "hidden_trans_audio_multimodal")
    odd = [k for k in missing if not ("cross_attn" in k or "prompt_embedding" in k or "pooler" in k or k.split(".")[0] in fresh)]
    assert not odd, odd[:8]
    assert torch.equal(model.state_dict()["audio_encoder.layer.3.attention.linears.1.weight"], mine["audio_encoder.layer.3.attention.linears.1.weight"])


@pytest.mark.parametrize("order", ["model_then_optimizer", "optimizer_then_model"])
def test_bf16_resume_keeps_fp32_masters_in_both_orders(order):
    """bf16 mode: the optimizer's fp32 masters carry bits the bf16 parameters do not. A resume restores them from the optimizer's own
    state dict; re-loading the model's (rounded) weights afterwards must not overwrite them (apex amp's master params survive
    `model.load_state_dict` the same way, apex/apex/amp/_process_optimizer.py:14-22), while genuinely different weights still re-sync."""
    from types import SimpleNamespace
    from valor_amd.model.valor import VALOR
    from valor_amd.optim import FusedAdamW
    spec = synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=3)
    opts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.01, betas=[0.9, 0.98])
    m0 = VALOR(None, spec=spec, dtype=torch.bfloat16, device="cpu")
    m0.load_state_dict(sd)
    o0 = FusedAdamW(m0, opts)
    o0.init_master_from(sd)                                  # masters = the full-precision weights: they differ from float(bf16 params)
    assert not torch.equal(o0.master, m0.arena.flat.float())
    ck_model = {k: v.clone() for k, v in m0.state_dict().items()}
    ck_opt = o0.state_dict()

    m1 = VALOR(None, spec=spec, dtype=torch.bfloat16, device="cpu")
    o1 = FusedAdamW(m1, opts)
    if order == "model_then_optimizer":
        m1.load_state_dict(ck_model)
        o1.load_state_dict(ck_opt)
    else:
        o1.load_state_dict(ck_opt)
        m1.load_state_dict(ck_model)
    assert torch.equal(o1.master, o0.master)                 # low-order bits survive either order
    assert torch.equal(m1.arena.flat, m0.arena.flat)
    # different weights loaded behind the optimizer's back still win over the stale masters
    sd2 = synth.make_state_dict(spec, seed=4)
    m1.load_state_dict(sd2)
    assert torch.equal(o1.master.to(torch.bfloat16), m1.arena.flat)
    changed = o1.master != o0.master
    assert torch.equal(o1.master[changed], m1.arena.flat.float()[changed])
