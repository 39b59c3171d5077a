"""MI355X-native VALOR pretraining model: host-side mirror of the reference interface
(model/pretrain.py::VALOR / model/modeling.py::VALORModel) whose arithmetic is entirely HIP kernels
(valor_amd.ops -> libvalor_hip.so).

Drop-in surface kept from the reference:
  * VALOR.from_pretrained(opts, state_dict)          (modeling.py:107-115, strict=False semantics)
  * VALOR.forward(batch, task, compute_loss=True)    (pretrain.py:125-134; 'pt_*' tasks -> forward_pt :214-541)
  * batch schema of valor_collate                    (data/data.py:423-428)
  * state_dict() / load_state_dict() key names       (utils/save.py:45-64 checkpoints)
Design differences (MI355X-first, results identical):
  * token-major [N, L, E] activations everywhere, fused QKV / KV projections from packed weights,
    flash attention reading the fused GEMM output in place;
  * cross-attention K/V of the concatenated [video | audio] tokens are projected ONCE per decoder
    layer and shared by all caption / mlm passes (the reference re-projects them in every pass,
    bert.py:316-317,450); the caption-tva/-tv/-ta passes run as ONE batched decoder pass whose
    query groups attend to different K/V row ranges;
  * pre-LN blocks fuse residual-add + next LayerNorm (+ dropout) in one kernel; post-LN BERT fuses
    bias + dropout + residual + LayerNorm;
  * masked-row gather indices and attention masks come from the host-side TokenMasker (no device
    round trip, modeling.py:134-174 runs on CPU in the reference too).
"""
import math
import os
import random

import numpy as np
import torch
from torch import nn

from .. import ops, streams
from ..arena import ParamArena
from ..hoststage import HostStage
from ..lib import ACT_GELU_ERF, ACT_QUICK_GELU, ACT_RELU
from ..synth import ValorSpec, base_spec, swin_relative_position_index, swin_spec, synthetic_vocab
from .params import optimizer_group, param_table

PROMPTS = {
    "caption": "describe the video with natural language",                     # pretrain.py:438
    "mlm_tva": "predict masked tokens with visual and audio cues",             # pretrain.py:492
    "mlm_tv": "predict masked tokens with visual cues",                        # pretrain.py:505
    "mlm_ta": "predict masked tokens with audio cues",                         # pretrain.py:516
    "contra": "project language in common space",                              # pretrain.py:256
    "qa": "answer the question",                                               # pretrain.py:1270
}


class TokenMasker:
    """Host-side BERT-style masking with the python-`random` draw ORDER of the reference (modeling.py:122-174): per row, one
    uniform per non-pad position j >= 1 (redrawn for the whole row until at least one position is masked), then per masked
    position in row-major order one uniform (80 % [MASK] / 10 % random token / 10 % keep) and, in the 10 % branch only, one
    random.choice over the vocabulary range; labels -1 elsewhere. The tokens are host tensors (they come from the collate
    function), so nothing touches the device; the per-element Python / numpy indexing of the reference's loops is gone -- rows
    are handled with list comprehensions over the positions that actually draw (b = 64: ~0.5 ms instead of ~6 ms per call)."""

    def __init__(self, mask_token, range_start, range_end):
        self.mask_token = mask_token
        self.range = [range_start, range_end]

    def __call__(self, tokens, mask_prob):
        tokens = np.array(tokens.cpu().numpy())
        labels = -np.ones(tokens.shape, dtype=np.int64)
        rnd = random.random
        picked = []
        for i in range(tokens.shape[0]):
            cand = (np.flatnonzero(tokens[i, 1:]) + 1).tolist()        # positions that draw: j >= 1 and not padding
            if not cand:
                raise ValueError("TokenMasker: a row without maskable tokens never terminates in the reference either")
            while True:
                sel = [j for j in cand if rnd() < mask_prob]
                if sel:
                    break
            picked.append(sel)
        choices = range(*self.range)         # random.choice(range) draws exactly like random.choice(list(range))
        for i, sel in enumerate(picked):
            row = tokens[i]
            for j in sel:
                src = int(row[j])
                prob = rnd()
                if prob < 0.8:
                    row[j] = self.mask_token
                elif prob < 0.9:
                    row[j] = random.choice(choices)
                labels[i, j] = src
        return torch.from_numpy(tokens).long(), torch.from_numpy(labels).long()


def _opt(opts, name, default):
    if opts is None:
        return default
    if isinstance(opts, dict):
        return opts.get(name, default)
    return getattr(opts, name, default)


class _DeferredKV:
    """the decoder's [video | audio] input, its K|V projections not issued yet (VALOR._graph_decoder)"""
    def __init__(self, va):
        self.va = va


class _BlockKV:
    """per-layer K|V of the video tokens and of the audio tokens, each projected by its OWN cross-attention block (cross_attn_type
    'va_parallel' / 'video_audio' / 'audio_video', bert.py:430-436): v / a = list of [b, S, 2E] per layer, or None without that modality.
    A query group is then a tuple of modality letters, not a key range."""
    def __init__(self, v=None, a=None):
        self.v, self.a = v, a

    def of(self, m):
        return self.v if m == "v" else self.a


class VALOR(nn.Module):
    def __init__(self, opts=None, spec: ValorSpec = None, dtype=torch.bfloat16, device="cuda", vocab_tokens=None):
        super().__init__()
        self.opts = opts
        # the two shipped base configurations: config/pretrain-VALOR-base.json (CLIP-ViT-B/16 video + CLIP text) and
        # scripts/pretrain.sh:3-8 (VideoSwin-B video + the shared BERT as text encoder)
        vtype = _opt(opts, "video_encoder_type", None)
        ttype = _opt(opts, "txt_encoder_type", None)
        want = None
        if vtype is not None or ttype is not None:
            vtype = vtype or "clip_vit_base_16"
            ttype = ttype or ("bert_base_uncased" if vtype.startswith("videoswin") else "clip_vit_base_16")
            shared_bert = ttype.startswith("bert") and _opt(opts, "share_txt_and_multimodal", True)
            if vtype.startswith("clip_vit_base") and ttype.startswith("clip_vit_base") and _opt(opts, "init_clip_head", True):
                want = ("clip", "clip")
            elif vtype.startswith("videoswin_base") and shared_bert:
                want = ("swin", "bert")
            elif vtype.startswith("clip_vit") and shared_bert:         # config/pretrain-VALOR-large.json:10-14
                want = ("clip", "bert")
            else:
                raise NotImplementedError("covered: clip_vit_base video + clip_vit_base text, videoswin_base or clip_vit_{base,large} video + "
                                          f"shared bert text; got video={vtype} text={ttype}")
        if spec is None:
            if want == ("clip", "bert"):
                from ..synth import clip_large_spec
                spec = clip_large_spec() if vtype.startswith("clip_vit_large_14") else ValorSpec(txt_encoder="bert")
                res = int(_opt(opts, "video_resolution", 224))
                if res != spec.resolution:
                    spec.resolution = res
                if int(_opt(opts, "contra_dim", 512)) != spec.cdim:
                    spec.contra_dim = int(_opt(opts, "contra_dim", 512))
            else:
                spec = swin_spec() if want == ("swin", "bert") else base_spec()
        elif want is not None and want != (spec.video_encoder, spec.txt_encoder):
            raise NotImplementedError(f"opts ask for {want} encoders, the spec describes {(spec.video_encoder, spec.txt_encoder)}")
        if (spec.video_encoder, spec.txt_encoder) not in (("clip", "clip"), ("swin", "bert"), ("clip", "bert")):
            raise NotImplementedError("the reference loads CLIP as a whole: video/text encoders come as clip+clip, swin+bert or clip+bert")
        # contra_type / late_fusion change the parameter set (pretrain.py:100-116): they live in the spec; opts may override it
        ct, lf = _opt(opts, "contra_type", spec.contra_type), bool(_opt(opts, "late_fusion", spec.late_fusion))
        if ct not in ("fine", "coarse") or _opt(opts, "caption_type", "unimlm") not in ("unimlm", "lm"):
            raise NotImplementedError("contra_type 'fine' / 'coarse' and caption_type 'unimlm' / 'lm' only")
        if (ct, lf) != (spec.contra_type, spec.late_fusion):
            import dataclasses
            spec = dataclasses.replace(spec, contra_type=ct, late_fusion=lf)
        self.caption_type = _opt(opts, "caption_type", "unimlm")        # pretrain.py:76
        self.label_smoothing = float(_opt(opts, "label_smoothing", 0.0))    # pretrain.py:72-74: the caption FINETUNE loss only (:839-840)
        self._smoothing = 0.0                                            # label smoothing of the decoder passes being issued
        # bert.py:430-436: one cross-attention block over [video | audio] ('va_concate', every shipped configuration) or a block per modality,
        # summed ('va_parallel') or applied one after the other ('video_audio', 'audio_video'): another parameter set, so it lives in the spec
        xt = _opt(opts, "cross_attn_type", spec.cross_attn_type)
        if xt not in ("va_concate", "va_parallel", "video_audio", "audio_video"):
            raise NotImplementedError(f"cross_attn_type {xt!r} (bert.py:430 asserts the same four)")
        if xt != spec.cross_attn_type:
            import dataclasses
            spec = dataclasses.replace(spec, cross_attn_type=xt)
        # pretrain.py:79: the caption rows become [tokens | as many [MASK]s], position L/2 + i predicts token i + 1 (the finetune losses; generation
        # ignores the flag like the reference's, :878-900)
        self.full_masker = bool(_opt(opts, "full_masker", False))
        if self.full_masker and _opt(opts, "caption_type", "unimlm") != "unimlm":
            # the reference still passes full_masker into the decoder's mask and positions when caption_type is 'lm' (pretrain.py:835,1276)
            # while its rows are NOT doubled there (:808, :1226 sit in the 'unimlm' branch): a combination no shipped config uses. Refused
            # loudly instead of dropping the flag silently.
            raise NotImplementedError("full_masker with caption_type='lm' is not supported (no shipped configuration combines them)")
        self._full_attn = False           # full_masker attention / positions for the decoder passes being issued (forward_cap / forward_qa)
        if _opt(opts, "fineweight_type", "one") == "none":
            raise NotImplementedError("fineweight_type='none' is a TypeError in the reference too (pretrain.py:330)")
        self.spec = spec
        self.drop_path = float(_opt(opts, "drop_path_rate", spec.swin_drop_path))   # videoswin.py:393 (active in training)
        self.dtype = dtype
        self.device = torch.device(device)
        self.use_task_prompt = bool(_opt(opts, "use_task_prompt", False))
        self.contra_loss_ratio = float(_opt(opts, "contra_loss_ratio", 1.0))
        self.beam_size = int(_opt(opts, "beam_size", 3))                      # train_utils.py:692, :635
        self.beam_size_qa = int(_opt(opts, "beam_size_qa", 1))                # train_utils.py:693
        self.max_generation_len = int(_opt(opts, "max_generation_len", 30))
        self.p_drop = float(_opt(opts, "dropout", 0.1))          # bert json / base_cfg 0.1, train_utils.py:617
        new_names = tuple(_opt(opts, "new_params_name", ()) or ())
        self.table = param_table(self.spec)
        entries = [(n, s, optimizer_group(refs[0], new_names)) for n, s, refs in self.table]
        self.arena = ParamArena(entries, dtype, self.device)
        self.P = self.arena.params
        for name, p in self.P.items():
            self.register_parameter(name.replace(".", "__"), p)
        self.vocab_tokens = vocab_tokens if vocab_tokens is not None else synthetic_vocab(self.spec.vocab)
        self.vocab = {t: i for i, t in enumerate(self.vocab_tokens)}
        self.bos_token, self.eos_token, self.text_mask_token = self.vocab["[CLS]"], self.vocab["[SEP]"], self.vocab["[MASK]"]
        self.text_masker = TokenMasker(self.text_mask_token, 106, self.spec.vocab)     # modeling.py:673
        self.reducer = None
        self.stage = HostStage(self.device)
        self._const = {}
        self.gather_fn = None          # set by valor_amd.dist for world_size > 1
        self.collect = None            # optional dict: intermediate tensors for parity tests
        # the reference's `checkpointing` option (modeling.py:573,583-586,607): every layer of the video encoder (CLIP resblocks / VideoSwin
        # blocks), of the AST, of the CLIP text tower and of the multimodal decoder's training path (bert.py:510-513) keeps its inputs only
        # and is run again in backward (ops.CheckpointFn).
        self.checkpointing = bool(_opt(opts, "checkpointing", False))
        self._graphs_on = False        # enable_graphs(): the encoders replay hipGraphs (valor_amd/graphs.py)
        self._graph_segs = {}
        # frozen_vision / frozen_multimodal (modeling.py:319-322, 675-680): requires_grad = False on the reference's parameter-name rules.
        # A frozen parameter's weight-gradient GEMMs are not issued (ops: needs_input_grad), it is never reported to the data-parallel
        # reducer and never becomes active in the optimizer (optim/adamw.py:62-63 skips p.grad is None: no update, no weight decay);
        # with frozen_vision nothing below the video encoder's output needs a gradient, so its whole backward pass disappears.
        self.frozen_vision = bool(_opt(opts, "frozen_vision", False))
        self.frozen_multimodal = bool(_opt(opts, "frozen_multimodal", False))
        self.frozen_names = set()
        for name, _shape, refs in self.table:
            if all(self._ref_is_frozen(r) for r in refs):
                self.P[name].requires_grad_(False)
                self.frozen_names.add(name)

    def _ref_is_frozen(self, ref):
        """the reference's freezing rules on one of ITS parameter names"""
        if self.frozen_vision and self.spec.video_encoder == "clip" and ref.startswith("clip_model."):       # modeling.py:319-322 (clip branch only)
            return "visual" in ref[len("clip_model."):]
        if self.frozen_multimodal:                                                                        # modeling.py:675-680
            if ref.startswith("cls."):
                return True
            if ref.startswith("multimodal_encoder."):
                k = ref[len("multimodal_encoder."):]
                if "encoder" in k and "cross" not in k:
                    return True
                if "embeddings" in k and any(j in k for j in ("embeddings.word_embeddings", "embeddings.position_embeddings",
                                                              "embeddings.token_type_embeddings", "embeddings.LayerNorm")):
                    return True
        return False

    # ------------------------------------------------------------------ checkpoint layout
    @classmethod
    def from_pretrained(cls, opts, state_dict, **kw):
        """modeling.py:107-115: construct (randomly initialised modules, see init_parameters), then load `state_dict` with
        strict=False. The reference's constructor also reads the CLIP / BERT / AST / VideoSwin component weights from
        ./pretrained_weights (modeling.py:512-660); here those come through `state_dict` (valor_amd.checkpoint.load_pretrained_components
        maps the component files to VALOR keys), and whatever `state_dict` does not cover keeps its initialisation. The keys that
        stayed at their initial value are kept in `model.missing_keys`."""
        model = cls(opts, **kw)
        model.init_parameters(seed=int(_opt(opts, "seed", 42)))
        model.missing_keys, model.unexpected_keys = [], []
        if state_dict:
            model.missing_keys, model.unexpected_keys = model.load_state_dict(state_dict, strict=False)
        else:
            model.missing_keys = [r for _, _, refs in model.table for r in refs if r != "cls.decoder.weight"]
        return model

    # nn.Linear modules the reference constructs OUTSIDE BertModel and never re-initialises: VALORModel.init_weights (modeling.py:92-105)
    # is defined but not applied anywhere; only BertModel applies init_bert_weights (bert.py:747). They keep torch's nn.Linear default.
    _TORCH_DEFAULT_LINEAR = ("contra_head_", "text_fine_weight.", "video_fine_weight.", "audio_fine_weight.", "va_fusion.", "hidden_trans_video_multimodal.0.",
                             "hidden_trans_audio_multimodal.0.", "cls.dense.")

    def init_parameters(self, seed=42):
        """Initialisation of whatever a checkpoint does not cover, following what the reference's constructor leaves behind:
        the multimodal BERT gets init_bert_weights (bert.py:617-630,747: Linear / Embedding weights N(0, 0.02), LayerNorm gain 1,
        biases 0); the nn.Linear modules outside it (Contra_head, the *_fine_weight MLPs, hidden_trans_*, cls.dense) keep torch's default
        -- weight and bias U(-1/sqrt(fan_in), 1/sqrt(fan_in)) -- because VALORModel.init_weights (modeling.py:92-105) is never applied;
        modeling.py:338-341 type / frame embeddings 0.02 N(0,1); pretrain.py:117 contra_temp 0.07; clip.py:295 logit_scale ln(1/0.07).
        Every rank seeds the same generator, so replicas start identical (TrainEngine broadcasts rank 0's arena besides). Component
        weights (CLIP, VideoSwin, AST, BERT) are pretrained in the reference; without their files they get N(0, 0.02) / 1 / 0."""
        from ..synth import state_dict_layout
        kinds = {k: kind for k, _, kind in state_dict_layout(self.spec)}
        gen = torch.Generator(device="cpu").manual_seed(int(seed))
        fan_in = {}
        for name, shape, refs in self.table:
            if refs[0].startswith(self._TORCH_DEFAULT_LINEAR) and refs[0].endswith(".weight"):
                fan_in[refs[0][:-len("weight")]] = int(shape[-1])
        with torch.no_grad():
            for name, shape, refs in self.table:
                kind = kinds.get(refs[0], "w")
                p = self.P[name]
                mod = refs[0].rsplit(".", 1)[0] + "."
                if mod in fan_in and kind in ("w", "b"):
                    bound = 1.0 / math.sqrt(fan_in[mod])
                    p.copy_(((2.0 * torch.rand(shape, generator=gen) - 1.0) * bound).to(p.dtype))
                elif kind == "g":
                    p.fill_(1.0)
                elif kind == "b":
                    p.zero_()
                elif kind == "s":
                    p.fill_(math.log(1 / 0.07) if "logit_scale" in name else 0.07)
                else:
                    p.copy_((0.02 * torch.randn(shape, generator=gen)).to(p.dtype))
        self._params_changed()

    def _params_changed(self):
        """parameters were rewritten behind the optimizer's back: refresh its fp32 masters (bf16 mode)"""
        for ref in list(getattr(self, "_optimizers", [])):
            opt = ref()
            if opt is not None:
                opt.sync_master()

    def zero_grad(self, set_to_none=False):
        """Gradients live in the flat arena and must stay bound to it (ops.GradSink, the reducer and the fused optimizer read the
        arena): zero in place and re-attach the views whatever `set_to_none` says."""
        self.arena.grad.zero_()
        self.arena.rebind_grads()

    def state_dict(self, *a, **k):
        """Reference-keyed state dict (fp32 CPU-agnostic views of the arena parameters)."""
        out = {}
        for name, shape, refs in self.table:
            p = self.P[name].detach()
            if len(refs) == 1 or refs[1] == "cls.decoder.weight":
                for r in refs:
                    out[r] = p
            else:
                rows = shape[0] // len(refs)
                for i, r in enumerate(refs):
                    out[r] = p[i * rows:(i + 1) * rows]
        if self.spec.video_encoder == "swin":
            # the integer buffer of every WindowAttention3D (videoswin.py:126) is part of the reference's state dict
            relidx = swin_relative_position_index(self.spec.swin_window)
            for name, _, _ in self.table:
                if name.endswith("attn.relative_position_bias_table"):
                    out[name.replace("relative_position_bias_table", "relative_position_index")] = relidx
        if self.spec.txt_encoder == "bert":
            # ... and so is the txt_encoder.* view of the shared multimodal encoder (modeling.py:689-691)
            for k in [k for k in out if k.startswith("multimodal_encoder.")]:
                out["txt_encoder." + k[len("multimodal_encoder."):]] = out[k]
        return out

    def load_state_dict(self, sd, strict=True):
        missing, used = [], set()
        with torch.no_grad():
            for name, shape, refs in self.table:
                p = self.P[name]
                if len(refs) == 1 or refs[1] == "cls.decoder.weight":
                    if refs[0] in sd:
                        p.copy_(sd[refs[0]].to(p.dtype).view(shape)); used.update(r for r in refs if r in sd)
                    else:
                        missing.append(refs[0])
                else:
                    rows = shape[0] // len(refs)
                    for i, r in enumerate(refs):
                        if r in sd:
                            p[i * rows:(i + 1) * rows].copy_(sd[r].to(p.dtype)); used.add(r)
                        else:
                            missing.append(r)
        unexpected = [k for k in sd if k not in used and not (
            (self.spec.video_encoder == "swin" and k.endswith("relative_position_index")) or
            (self.spec.txt_encoder == "bert" and k.startswith("txt_encoder.") and "multimodal_encoder." + k[12:] in used))]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        self._params_changed()
        return missing, unexpected

    def ref_named_groups(self):
        """(internal name, optimizer group id) -- used by valor_amd.optim."""
        return {n: self.arena.groups[n] for n, _, _ in self.table}

    # ------------------------------------------------------------------ host helpers
    def get_task_prompt(self, sentence, batch_size):
        """modeling.py:355-369 (bert tokenizer branch): whole-word / greedy WordPiece lookup."""
        ids = [self.bos_token]
        for wd in sentence.lower().split():
            if wd in self.vocab:
                ids.append(self.vocab[wd]); continue
            start, pieces, bad = 0, [], False
            while start < len(wd):
                end, cur = len(wd), None
                while start < end:
                    sub = wd[start:end] if start == 0 else "##" + wd[start:end]
                    if sub in self.vocab:
                        cur = sub; break
                    end -= 1
                if cur is None:
                    bad = True; break
                pieces.append(self.vocab[cur]); start = end
            ids.extend([self.vocab.get("[UNK]", 100)] if bad else pieces)
        ids.append(self.eos_token)
        return torch.tensor(ids, dtype=torch.long).unsqueeze(0).expand(batch_size, -1).contiguous()

    @staticmethod
    def _bert_mask(tokens_cpu, prompt_cpu, casual, full_masker=False):
        """additive attention mask of BertModel.forward, bert.py:854-885 -> fp32 [B, T, T].
        numpy on purpose: these are tiny tensors and a torch CPU op would wake the whole intra-op thread pool."""
        am = (tokens_cpu.numpy() != 0).astype(np.int64)
        token_len = am.shape[1]
        if prompt_cpu is not None:
            am = np.concatenate((am, (prompt_cpu.numpy() != 0).astype(np.int64)), axis=1)
        total = am.shape[1]
        am = np.repeat(am[:, None, :], total, axis=1)
        if casual and full_masker:                                   # bert.py:872-878
            n = token_len // 2
            am[:, :n, :n] = np.tril(am[:, :n, :n])
            am[:, :n, n:token_len] = 0
            am[:, n:token_len, :n] = np.tril(am[:, n:token_len, :n])
            am[:, n:token_len, n:token_len] = np.eye(n, dtype=am.dtype)
            am[:, token_len:, :token_len] = 0
        elif casual:
            am[:, :token_len, :token_len] = np.tril(am[:, :token_len, :token_len])
            am[:, token_len:, :token_len] = 0
        return torch.from_numpy(((1.0 - am.astype(np.float32)) * -10000.0).astype(np.float32))

    @staticmethod
    def _clip_text_mask(tokens_cpu):
        """clip.py:382-414 with casual=True -> fp32 [B, L, L]"""
        m = (tokens_cpu.numpy() != 0).astype(np.int64)
        L = m.shape[1]
        m = np.tril(np.repeat(m[:, None, :], L, axis=1))
        return torch.from_numpy(((1.0 - m.astype(np.float32)) * -10000.0).astype(np.float32))

    def _dev(self, t, dtype=None):
        """host tensor -> device through the pinned staging ring (asynchronous); device tensors pass through."""
        return self.stage.put(t, dtype)

    def _const_idx(self, n, stride):
        """cached device tensor arange(n) * stride (cls-token row indices)"""
        key = (n, stride)
        t = self._const.get(key)
        if t is None:
            t = (torch.arange(n) * stride).to(self.device)
            self._const[key] = t
        return t

    # ------------------------------------------------------------------ encoders
    def _clip_blocks(self, x, prefix, n_layers, heads, mask, final_g, final_b):
        """pre-LN CLIP transformer (clip.py:194-214) + final LayerNorm; x: [N, L, E] residual stream."""
        P = self.P
        x, y = ops.layer_norm_stream(x, P[f"{prefix}.resblocks.0.ln_1.weight"], P[f"{prefix}.resblocks.0.ln_1.bias"], 1e-5)

        def block(i, x, y):
            """resblock i on (residual stream x, its LayerNorm y) -> the same pair for block i + 1 (the last one: the final LayerNorm only)"""
            p = f"{prefix}.resblocks.{i}."
            qkv = ops.linear(y, P[p + "attn.in_proj_weight"], P[p + "attn.in_proj_bias"])
            a = ops.self_attention(qkv, heads, mask, 0.0)
            o = ops.linear(a, P[p + "attn.out_proj.weight"], None)
            x, y = ops.bias_dropout_residual_ln(o, P[p + "attn.out_proj.bias"], x, P[p + "ln_2.weight"], P[p + "ln_2.bias"], 1e-5, 0.0, True)
            # the c_proj bias is added by the fused residual+LayerNorm kernel (its gradient then falls out of that kernel's
            # backward for free instead of a separate column-sum pass over dY)
            m = ops.mlp(y, P[p + "mlp.c_fc.weight"], P[p + "mlp.c_fc.bias"], P[p + "mlp.c_proj.weight"], None, ACT_QUICK_GELU)
            if i + 1 < n_layers:
                q = f"{prefix}.resblocks.{i + 1}."
                return ops.bias_dropout_residual_ln(m, P[p + "mlp.c_proj.bias"], x, P[q + "ln_1.weight"], P[q + "ln_1.bias"], 1e-5, 0.0, True)
            return ops.bias_dropout_residual_ln(m, P[p + "mlp.c_proj.bias"], x, final_g, final_b, 1e-5, 0.0, False)

        ckpt = self.checkpointing and self.training and torch.is_grad_enabled()
        for i in range(n_layers):
            out = ops.checkpoint(lambda x_, y_, i=i: block(i, x_, y_), x, y) if ckpt else block(i, x, y)
            if i + 1 < n_layers:
                x, y = out
            else:
                y = out
        return y

    # ------------------------------------------------------------------ VideoSwin
    def _swin_geometry(self, D, H, W, shifted):
        """Device-resident index maps of one block geometry (cached): roll + window_partition as a row map, the shift mask as
        per-slot region labels (compute_mask, videoswin.py:272-285), rel[] of the relative position bias (:112-126)."""
        key = ("swin", D, H, W, shifted)
        geo = self._const.get(key)
        if geo is not None:
            return geo
        full = self.spec.swin_window
        win = tuple(s if s <= w else w for s, w in zip((D, H, W), full))                  # get_window_size :86-99, on the UNPADDED map
        sh = tuple(0 if (s <= w or not shifted) else w // 2 for s, w in zip((D, H, W), full))
        # zero padding up to whole windows (videoswin.py:198-203; e.g. 10 or 12 test frames against the 8-deep window): roll, partition and
        # the shift mask (BasicLayer.forward :335-338) all live on the padded map
        D0, H0, W0 = D, H, W
        D, H, W = (-(-s // w) * w for s, w in zip((D, H, W), win))
        wd, wh, ww = win
        part = lambda a: a.reshape(D // wd, wd, H // wh, wh, W // ww, ww).transpose(0, 2, 4, 1, 3, 5).reshape(-1)
        idx = np.arange(D * H * W).reshape(D, H, W)
        label = None
        if any(sh):
            idx = np.roll(idx, (-sh[0], -sh[1], -sh[2]), axis=(0, 1, 2))                  # shifted_x[pos] = x[pos + shift]
            ids = []
            for X, w, s_ in zip((D, H, W), win, sh):
                r = np.zeros(X, dtype=np.int64)
                if s_ > 0:
                    r[X - w:X - s_] = 1; r[X - s_:] = 2
                else:
                    r[:] = 2
                ids.append(r)
            lab = ids[0][:, None, None] * 9 + ids[1][None, :, None] * 3 + ids[2][None, None, :]
            label = torch.from_numpy(part(lab).astype(np.uint8)).to(self.device)
        N = wd * wh * ww
        fd, fh, fw = np.meshgrid(np.arange(full[0]), np.arange(full[1]), np.arange(full[2]), indexing="ij")
        lin = (fd * (2 * full[1] - 1) * (2 * full[2] - 1) + fh * (2 * full[2] - 1) + fw).reshape(-1)
        inv = -np.ones(int(lin[-1]) + 1, dtype=np.int32)
        inv[lin[:N]] = np.arange(N, dtype=np.int32)                                          # slot whose rel is m (bias-table gradient)
        geo = dict(rowmap=torch.from_numpy(part(idx).astype(np.int32)).to(self.device), label=label,
                   rel=torch.from_numpy(lin[:N].astype(np.int32)).to(self.device), relc=int(lin[-1]),
                   rel_inv=torch.from_numpy(inv).to(self.device),
                   nW=(D // wd) * (H // wh) * (W // ww), N=N, rows=D * H * W, padded=(D, H, W) != (D0, H0, W0))
        if geo["padded"]:               # natural row of every padded position (-1: a zero row) and back, per sample
            nat = -np.ones((D, H, W), dtype=np.int64)
            nat[:D0, :H0, :W0] = np.arange(D0 * H0 * W0).reshape(D0, H0, W0)
            geo["pad_src"] = nat.reshape(-1)
            geo["unpad_src"] = np.arange(D * H * W).reshape(D, H, W)[:D0, :H0, :W0].reshape(-1)
            geo["rows0"] = D0 * H0 * W0
        self._const[key] = geo
        return geo

    def _swin_pad_idx(self, geo, b):
        """(pad, unpad) gather indices of a padded block geometry for a batch of b clips: natural rows -> padded map (zero rows at -1),
        padded attention output -> natural rows (the crop x[:, :D, :H, :W], videoswin.py:222-223)"""
        key = ("swinpad", id(geo), b)
        t = self._const.get(key)
        if t is None:
            ps, us, r0, rp = geo["pad_src"], geo["unpad_src"], geo["rows0"], geo["rows"]
            pad = np.concatenate([np.where(ps >= 0, ps + i * r0, -1) for i in range(b)])
            unpad = np.concatenate([us + i * rp for i in range(b)])
            t = (torch.from_numpy(pad).to(self.device), torch.from_numpy(unpad).to(self.device))
            self._const[key] = t
        return t

    def _swin_merge_idx(self, b, D, H, W):
        """PatchMerging's 2x2 neighbour gather (videoswin.py:262-266) as source rows of the [.., 4C] output viewed as 4 C-rows;
        odd H / W are zero padded first (:257-259): index -1 = a zero row"""
        key = ("merge", b, D, H, W)
        t = self._const.get(key)
        if t is None:
            Hp, Wp = H + H % 2, W + W % 2
            g = -np.ones((b, D, Hp, Wp), dtype=np.int64)
            g[:, :, :H, :W] = np.arange(b * D * H * W).reshape(b, D, H, W)
            q = np.stack([g[:, :, 0::2, 0::2], g[:, :, 1::2, 0::2], g[:, :, 0::2, 1::2], g[:, :, 1::2, 1::2]], axis=-1)
            t = torch.from_numpy(q.reshape(-1).astype(np.int64)).to(self.device)
            self._const[key] = t
        return t

    def forward_video_encoder_swin(self, video_pixels):
        """modeling.py:452-455 -> SwinTransformer3D.forward videoswin.py:441-458. Returns [b, F, H/32 * W/32, C_out].
        Tokens stay in natural [b, D, H, W] row order through the whole encoder: roll / window_partition / window_reverse
        live in the attention kernel's index map, PatchMerging is one row gather."""
        sp = self.spec
        b = video_pixels.shape[0]
        vid = self._dev(video_pixels.float().contiguous())
        total = sum(sp.swin_depths)
        rate = self.drop_path if self.training else 0.0
        scales = None
        if rate > 0:                    # stochastic depth, videoswin.py:40-49: one per-sample keep decision per residual branch
            keep = 1.0 - np.linspace(0.0, rate, total)                                          # dpr, videoswin.py:418
            u = np.random.random_sample((total, 2, b))
            scales = self._dev(torch.from_numpy((np.floor(keep[:, None, None] + u) / keep[:, None, None]).astype(np.float32)))
        if self._use_graphs():
            # the keep decisions are drawn on the host (numpy, like the reference's torch.rand per block) and enter the graph as an INPUT
            name = "swin" if scales is None else "swin_droppath"
            seg = self._graph_segs.get(name)
            if seg is None:
                from .. import graphs
                seg = self._graph_segs[name] = graphs.GraphedSegment(name, self._video_encoder_swin)
            return seg(vid) if scales is None else seg(vid, scales)
        return self._video_encoder_swin(vid, scales)

    def _video_encoder_swin(self, vid, scales=None):
        """vid: [b, F, 3, H, W] fp32 on the device; scales: fp32 [blocks, 2, b] stochastic-depth factors or None -> [b, D, H/32 * W/32, C_out]"""
        P, sp = self.P, self.spec
        b, F, c, h, w = vid.shape
        C = sp.swin_embed
        e = "video_encoder.patch_embed."
        tok = ops.linear(ops.patchify3d(vid, 4, self.dtype), ops.param_view(P[e + "proj.weight"], C, -1), P[e + "proj.bias"])
        D, H, W = F, h // 4, w // 4
        x = ops.layer_norm(tok, P[e + "norm.weight"], P[e + "norm.bias"], 1e-5)                 # patch_norm; pos_drop p = 0
        k = 0
        ckpt = self.checkpointing and self.training and torch.is_grad_enabled()          # videoswin.py:234-241, 448-449
        n_stages = len(sp.swin_depths)

        def block(li, bi, k, depth, heads, D, H, W, x, y):
            """block bi of stage li on (residual stream x, norm1(x) = y) -> (x, y) for the next block of the stage, x alone at the end of a
            stage that is followed by PatchMerging, the final LayerNorm's output at the end of the last stage"""
            rows = D * H * W
            p = f"video_encoder.layers.{li}.blocks.{bi}."
            s1 = scales[k, 0] if (scales is not None and k > 0) else None
            s2 = scales[k, 1] if (scales is not None and k > 0) else None
            geo = self._swin_geometry(D, H, W, bi % 2 == 1)
            if geo["padded"]:       # zero rows AFTER norm1 (videoswin.py:196-202): their q / k / v are the bias, they are attended to unmasked
                pad, unpad = self._swin_pad_idx(geo, b)
                qkv = ops.linear(ops.gather_rows(y, pad), P[p + "attn.qkv.weight"], P[p + "attn.qkv.bias"])
                a = ops.gather_rows(ops.window_attention(qkv, P[p + "attn.relative_position_bias_table"], geo, heads, b), unpad)
            else:
                qkv = ops.linear(y, P[p + "attn.qkv.weight"], P[p + "attn.qkv.bias"])
                a = ops.window_attention(qkv, P[p + "attn.relative_position_bias_table"], geo, heads, b)
            o = ops.linear(a, P[p + "attn.proj.weight"], None)
            x, y2 = ops.bias_dropout_residual_ln(o, P[p + "attn.proj.bias"], x, P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5, 0.0, True, s1, rows)
            m = ops.mlp(y2, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"], P[p + "mlp.fc2.weight"], None, ACT_GELU_ERF)
            if bi + 1 < depth:
                q = f"video_encoder.layers.{li}.blocks.{bi + 1}."
                return ops.bias_dropout_residual_ln(m, P[p + "mlp.fc2.bias"], x, P[q + "norm1.weight"], P[q + "norm1.bias"], 1e-5, 0.0, True, s2, rows)
            if li + 1 < n_stages:
                return ops.bias_dropout_residual(m, P[p + "mlp.fc2.bias"], x, 0.0, s2, rows)
            return ops.bias_dropout_residual_ln(m, P[p + "mlp.fc2.bias"], x, P["video_encoder.norm.weight"], P["video_encoder.norm.bias"], 1e-5, 0.0, False, s2, rows)

        for li, (depth, heads) in enumerate(zip(sp.swin_depths, sp.swin_heads)):
            p0 = f"video_encoder.layers.{li}.blocks.0."
            y = ops.layer_norm(x, P[p0 + "norm1.weight"], P[p0 + "norm1.bias"], 1e-5)
            for bi in range(depth):
                fn = lambda x_, y_, li=li, bi=bi, k=k, depth=depth, heads=heads, D=D, H=H, W=W: block(li, bi, k, depth, heads, D, H, W, x_, y_)
                out = ops.checkpoint(fn, x, y) if ckpt else fn(x, y)
                if bi + 1 < depth:
                    x, y = out
                elif li + 1 < n_stages:
                    x = out
                else:
                    y = out
                k += 1
            if li + 1 < len(sp.swin_depths):
                d = f"video_encoder.layers.{li}.downsample."
                xm = ops.gather_rows(x, self._swin_merge_idx(b, D, H, W)).view(-1, 4 * C)
                xm = ops.layer_norm(xm, P[d + "norm.weight"], P[d + "norm.bias"], 1e-5)
                x = ops.linear(xm, P[d + "reduction.weight"], None)
                C, H, W = 2 * C, (H + 1) // 2, (W + 1) // 2
        return y.view(b, D, H * W, C)

    def forward_txt_encoder_bert(self, bert_tokens_cpu):
        """modeling.py:439-440: the shared multimodal BERT run on the text alone (no cross-attention input), casual=False;
        pretrain.py:254-263 appends the task prompt when use_task_prompt and drops its rows again. Returns [b, L, hidden]."""
        L = bert_tokens_cpu.shape[1]
        prompt = self.get_task_prompt(PROMPTS["contra"], bert_tokens_cpu.shape[0]) if self.use_task_prompt else None
        x = self._bert_embed(self._dev(bert_tokens_cpu), L, None)
        if prompt is not None:
            x = torch.cat((x, self._bert_embed(self._dev(prompt), prompt.shape[1], "prompt")), dim=1)
        mask = self._dev(self._bert_mask(bert_tokens_cpu, prompt, False))
        out = self.bert_encoder(x, mask, None, None, 0)
        return out[:, :L].contiguous() if prompt is not None else out

    def forward_video_encoder(self, video_pixels):
        """modeling.py:449-465 (clip) -> VisionTransformer.forward clip.py:259-274. Returns [b, F, 197, W]."""
        if self.spec.video_encoder == "swin":
            return self.forward_video_encoder_swin(video_pixels)
        P, sp = self.P, self.spec
        b, n, c, h, w = video_pixels.shape
        imgs = self._dev(video_pixels.reshape(b * n, c, h, w).float())
        if self._use_graphs() and not self.frozen_vision:      # (a frozen tower has no backward pass to capture)
            seg = self._graph_segs.get("vit")
            if seg is None:
                from .. import graphs
                seg = self._graph_segs["vit"] = graphs.GraphedSegment("vit", self._video_encoder_clip)
            return seg(imgs).view(b, n, sp.vis_tokens, sp.vis_width)
        return self._video_encoder_clip(imgs).view(b, n, sp.vis_tokens, sp.vis_width)

    def _use_graphs(self):
        """hipGraph replay of the encoders (valor_amd/graphs.py): training steps only, on a GPU, once enable_graphs() was called"""
        return self._graphs_on and self.training and torch.is_grad_enabled() and self.device.type == "cuda" and self.collect is None

    def enable_graphs(self, on=True):
        """capture the CLIP ViT and AST encoders' forward + backward as hipGraphs from their third training call on (per input shape).
        Switches ops.DropoutState to device mode (by-value offsets restart every step, a device-resident per-step counter is added in
        the kernels): whoever drives the steps calls ops.DropoutState.begin_step() at the top of each (TrainEngine does)."""
        self._graphs_on = bool(on)
        if on:
            ops.DropoutState.enable_device_base(self.device)
        else:
            for seg in self._graph_segs.values():
                seg.release()
            self._graph_segs = {}
            from .. import graphs
            graphs.release_all()
            ops.DropoutState.disable_device_base()        # back to host mode: by-value windows only, nothing left set for a later model / device

    def _video_encoder_clip(self, imgs):
        """imgs: [b * F, 3, H, W] fp32 on the device -> [b * F, tokens, width]"""
        P, sp = self.P, self.spec
        bn = imgs.shape[0]
        wconv = ops.param_view(P["clip_model.visual.conv1.weight"], sp.vis_width, -1)
        vec = 8 if self.dtype == torch.bfloat16 else 4
        kp = (wconv.shape[1] + vec - 1) // vec * vec
        # ViT-L/14: 3 * 14 * 14 = 588 contraction elements per patch; the GEMM stages 16-byte chunks, so operand rows are
        # zero-padded to the next chunk (592). Pure data movement; the weight's gradient flows back through the column slice.
        patches = ops.patchify(imgs, sp.patch, self.dtype, pad_to=kp)
        if kp != wconv.shape[1]:
            wconv = ops.pad_cols(wconv, kp)
        tok = ops.linear(patches, wconv, None)
        Pn = sp.vis_tokens - 1
        x = ops.assemble_tokens(tok, P["clip_model.visual.class_embedding"], P["clip_model.visual.positional_embedding"], None, bn, Pn)
        x = ops.layer_norm(x, P["clip_model.visual.ln_pre.weight"], P["clip_model.visual.ln_pre.bias"], 1e-5)
        return self._clip_blocks(x, "clip_model.visual.transformer", sp.vis_layers, sp.vis_heads, None,
                                 P["clip_model.visual.ln_post.weight"], P["clip_model.visual.ln_post.bias"])

    def forward_txt_encoder(self, clip_tokens_cpu):
        """modeling.py:437-446 -> CLIP.encode_text(casual=True) clip.py:372-427. Returns [b, L, TW]."""
        ids = self._dev(clip_tokens_cpu)
        mask = self._dev(self._clip_text_mask(clip_tokens_cpu))
        if self._use_graphs():
            seg = self._graph_segs.get("clip_text")
            if seg is None:
                from .. import graphs
                seg = self._graph_segs["clip_text"] = graphs.GraphedSegment("clip_text", self._txt_encoder_clip)
            return seg(ids, mask)
        return self._txt_encoder_clip(ids, mask)

    def _txt_encoder_clip(self, ids, mask):
        """ids [b, L] token ids, mask [b, L, L] additive attention mask, both on the device -> [b, L, TW]"""
        P, sp = self.P, self.spec
        x = ops.embed(ids, P["clip_model.token_embedding.weight"], P["clip_model.positional_embedding"], None, ids.shape[1])
        return self._clip_blocks(x, "clip_model.transformer", sp.txt_layers, sp.txt_heads, mask,
                                 P["clip_model.ln_final.weight"], P["clip_model.ln_final.bias"])

    def forward_audio_encoder(self, audio):
        """modeling.py:468-480; AudioEmbeddings :750-762; pre-LN TransformerEncoder transformer.py:74-85,156-170.
        Returns [b, A, 129, AW]."""
        sp = self.spec
        b, n, hh, ww = audio.shape
        spec_in = self._dev(audio.reshape(b * n, 1, hh, ww).float())
        if self._use_graphs():
            seg = self._graph_segs.get("ast")
            if seg is None:
                from .. import graphs
                seg = self._graph_segs["ast"] = graphs.GraphedSegment("ast", self._audio_encoder)
            return seg(spec_in).view(b, n, sp.aud_tokens, sp.aud_width)
        return self._audio_encoder(spec_in).view(b, n, sp.aud_tokens, sp.aud_width)

    def _audio_encoder(self, spec_in):
        """spec_in: [b * A, 1, melbins, frames] fp32 on the device -> [b * A, tokens, width]"""
        P, sp, p = self.P, self.spec, (self.p_drop if self.training else 0.0)
        bn = spec_in.shape[0]
        patches = ops.patchify(spec_in, sp.aud_patch, self.dtype)
        tok = ops.linear(patches, ops.param_view(P["audio_embeddings.first_conv.weight"], sp.aud_width, -1), None)
        Pn = sp.aud_tokens - 1
        x = ops.assemble_tokens(tok, P["audio_embeddings.cls_token"], P["audio_embeddings.position_embeddings.weight"],
                                P["audio_embeddings.first_conv.bias"], bn, Pn)
        if p > 0:
            x = ops.bias_dropout_residual(x, None, None, p)
        x, y = ops.layer_norm_stream(x, P["audio_encoder.layer.0.layernorm1.weight"], P["audio_encoder.layer.0.layernorm1.bias"], 1e-12)

        def layer(i, x, y):
            q = f"audio_encoder.layer.{i}."
            qkv = ops.linear(y, P[q + "attention.qkv.weight"], P[q + "attention.qkv.bias"])
            a = ops.self_attention(qkv, sp.aud_heads, None, p)
            o = ops.linear(a, P[q + "attention.linears.3.weight"], None)
            x, y = ops.bias_dropout_residual_ln(o, P[q + "attention.linears.3.bias"], x, P[q + "layernorm2.weight"], P[q + "layernorm2.bias"], 1e-12, p, True)
            m = ops.mlp(y, P[q + "ff_layer.linear1.weight"], P[q + "ff_layer.linear1.bias"], P[q + "ff_layer.linear2.weight"], None, ACT_GELU_ERF)
            b2 = P[q + "ff_layer.linear2.bias"]
            if i + 1 < sp.aud_layers:
                r = f"audio_encoder.layer.{i + 1}."
                return ops.bias_dropout_residual_ln(m, b2, x, P[r + "layernorm1.weight"], P[r + "layernorm1.bias"], 1e-12, p, True)
            return ops.bias_dropout_residual_ln(m, b2, x, P["audio_encoder.last_layernorm.weight"], P["audio_encoder.last_layernorm.bias"], 1e-12, p, False)

        ckpt = self.checkpointing and self.training and torch.is_grad_enabled()          # transformer.py:163-164
        for i in range(sp.aud_layers):
            out = ops.checkpoint(lambda x_, y_, i=i: layer(i, x_, y_), x, y) if ckpt else layer(i, x, y)
            if i + 1 < sp.aud_layers:
                x, y = out
            else:
                y = out
        return y

    # ------------------------------------------------------------------ multimodal decoder
    def _bert_embed(self, ids_dev, L, token_type, full_masker=False):
        """BertEmbeddings.forward bert.py:190-218: word + position + (token_type[0] | prompt) -> LN -> dropout.
        full_masker (:197-201): the second half of the row sits at positions 1 .. L/2 -- two lookups, the second one into the position
        table from its row 1 on."""
        P, e = self.P, "multimodal_encoder.embeddings."
        tv = P[e + "prompt_embedding.weight"][0] if token_type == "prompt" else P[e + "token_type_embeddings.weight"][0]
        if full_masker and token_type is None:
            n = L // 2
            ids2 = ids_dev.view(-1, L)
            xa = ops.embed(ids2[:, :n].contiguous(), P[e + "word_embeddings.weight"], P[e + "position_embeddings.weight"], tv, n)
            xb = ops.embed(ids2[:, n:].contiguous(), P[e + "word_embeddings.weight"], P[e + "position_embeddings.weight"][1:], tv, n)
            x = torch.cat((xa.view(ids2.shape[0], n, -1), xb.view(ids2.shape[0], n, -1)), dim=1)
        else:
            x = ops.embed(ids_dev, P[e + "word_embeddings.weight"], P[e + "position_embeddings.weight"], tv, L)
        x = ops.layer_norm(x, P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], 1e-12)
        p = self.p_drop if self.training else 0.0
        if p > 0:
            x = ops.bias_dropout_residual(x, None, None, p)
        return x

    def project_cross_kv(self, va_input):
        """K|V of the concatenated [video | audio] tokens, ONCE per decoder layer (shared by every pass). The 12 projections
        read the same input: their input gradients meet in one buffer (ops.GradSlot); each layer's K|V buffer gets a slot
        of its own for the decoder passes that attend to it."""
        P = self.P
        va_slot = ops.GradSlot()
        self._kv_slots = [ops.GradSlot() for _ in range(self.spec.layers)]
        proj = lambda x, i: ops.linear(x, P[f"multimodal_encoder.encoder.layer.{i}.cross_attn.cross.kv.weight"],
                                       P[f"multimodal_encoder.encoder.layer.{i}.cross_attn.cross.kv.bias"], grad_slot=va_slot)
        self._dkv_static = None
        if not (streams.enabled() and self.device.type == "cuda" and os.environ.get("VALOR_KV_STREAM", "1") == "1" and self.training
                and torch.is_grad_enabled()):
            return [proj(va_input, i) for i in range(self.spec.layers)]
        # The 12 projections are chip-filling GEMMs (117 k rows), the decoder layers that consume them are not (8.8 k rows): on the side
        # stream they run BESIDE the decoder layers -- layer i waits for its own projection only -- and in backward each layer's
        # dgrad / wgrad pair runs there as soon as that layer's dK|dV is complete, beside the lower layers' backward (streams.py).
        # The K|V tensors and their gradients cross streams; handed through the caching allocator they could not be reused until the
        # other stream had passed them (round 2: the pool crept 62 -> 71 GB over 20 steps, a hipMalloc every other step). They live in
        # STATIC per-layer buffers instead (2 x 12 x 360 MB at the bench shape), reused every step: the side stream's first act of a
        # step is to wait for the main stream (fork), i.e. for every consumer of the previous step.
        side = streams.side_stream(self.device)
        shape = tuple(va_input.shape[:-1]) + (2 * self.spec.hidden,)
        key = (shape, va_input.dtype)
        # Contract: ONE training forward per backward. The buffers are rewritten by the next forward; every autograd node that saved one
        # remembers the generation it saw (ops.StaticGen) and raises instead of differentiating through a newer one. They are released by
        # eval() / a shape change (release_static_kv).
        if getattr(self, "_kv_static_key", None) != key:
            self.release_static_kv()
            with torch.cuda.stream(side):
                self._kv_static = [torch.empty(shape, dtype=va_input.dtype, device=self.device) for _ in range(self.spec.layers)]
                self._dkv_static_pool = [torch.empty(shape, dtype=va_input.dtype, device=self.device) for _ in range(self.spec.layers)]
            self._kv_static_key = key
            self._kv_gen = ops.StaticGen()
            self._kv_gen.register(self._kv_static)
        self._kv_gen.gen += 1
        self._dkv_static = self._dkv_static_pool
        va_s = streams.fork(side, va_input)
        items = []
        with torch.cuda.stream(side):
            for i in range(self.spec.layers):
                kv = ops.linear(va_s, P[f"multimodal_encoder.encoder.layer.{i}.cross_attn.cross.kv.weight"],
                                P[f"multimodal_encoder.encoder.layer.{i}.cross_attn.cross.kv.bias"], grad_slot=va_slot, out=self._kv_static[i])
                ev = torch.cuda.Event()
                ev.record(side)
                items.append((kv, ev))
        return streams.LazyTensors(side, items)

    def release_static_kv(self):
        """drop the static K|V / dK|dV pools of project_cross_kv (2 x layers x |K|V|: 8.6 GB at the base bench shape)"""
        g = getattr(self, "_kv_gen", None)
        if g is not None:
            g.gen += 1            # any graph still holding them must not run backward
            g.release()
        # ... and a CAPTURED decoder stack has their addresses baked in: its graphs go with the buffers (the next training steps run eagerly
        # again and re-capture on new buffers)
        seg = getattr(self, "_graph_segs", {}).get("decoder")
        if seg is not None:
            seg.release()
        self._kv_static = self._dkv_static_pool = self._dkv_static = self._kv_gen = None
        self._kv_static_key = None

    def train(self, mode=True):
        if not mode:
            self.release_static_kv()
        return super().train(mode)

    def cross_inputs(self, video_output, audio_output, defer_kv=False):
        """get_multimodal_forward_input_video / _audio (modeling.py:485-502) + the K|V projections of every decoder layer.
        Returns (kv_layers, {group: (first key row, rows)}). defer_kv: the caller is the training path that may run the layer stack as a
        graphed segment (_decoder_fused): the projections are then issued inside it."""
        P, sp = self.P, self.spec
        kv_layers, ranges = None, {}
        if video_output is not None or audio_output is not None:
            Sv = video_output.shape[1] * video_output.shape[2] if video_output is not None else 0
            Sa = audio_output.shape[1] * audio_output.shape[2] if audio_output is not None else 0
            if video_output is not None and "hidden_trans_video_multimodal.0.weight" in P:       # modeling.py:348-349,487-488
                hv = ops.linear(video_output.reshape(-1, sp.video_dim), P["hidden_trans_video_multimodal.0.weight"],
                                P["hidden_trans_video_multimodal.0.bias"])
                hv = ops.layer_norm(hv, P["hidden_trans_video_multimodal.1.weight"], P["hidden_trans_video_multimodal.1.bias"], 1e-12)
                video_output = hv.view(*video_output.shape[:3], sp.hidden)
            if audio_output is not None and "hidden_trans_audio_multimodal.0.weight" in P:       # modeling.py:350-351,497-498
                ha = ops.linear(audio_output.reshape(-1, sp.aud_width), P["hidden_trans_audio_multimodal.0.weight"],
                                P["hidden_trans_audio_multimodal.0.bias"])
                ha = ops.layer_norm(ha, P["hidden_trans_audio_multimodal.1.weight"], P["hidden_trans_audio_multimodal.1.bias"], 1e-12)
                audio_output = ha.view(*audio_output.shape[:3], sp.hidden)
            if sp.cross_attn_type != "va_concate":
                # a block per modality: each projects its own modality's tokens (own key / value weights). Plain tensors through autograd
                # (no static buffers, no graphed decoder): these modes run the per-pass decoder path (_decoder_groups)
                def project(x, fe, te, blk):
                    inp, slot = ops.single_input(x, P[fe], P[te]), ops.GradSlot()
                    return [ops.linear(inp, P[f"multimodal_encoder.encoder.layer.{i}.{blk}.cross.kv.weight"],
                                       P[f"multimodal_encoder.encoder.layer.{i}.{blk}.cross.kv.bias"], grad_slot=slot) for i in range(sp.layers)]
                kv = _BlockKV(project(video_output, "video_frame_embedding", "video_type_embeddings", "cross_attn_v") if video_output is not None else None,
                              project(audio_output, "audio_frame_embedding", "audio_type_embeddings", "cross_attn_a") if audio_output is not None else None)
                ranges = {}
                if kv.v is not None and kv.a is not None:
                    ranges["tva"] = ("v", "a")
                if kv.v is not None:
                    ranges["tv"] = ("v",)
                if kv.a is not None:
                    ranges["ta"] = ("a",)
                return kv, ranges
            if video_output is not None and audio_output is not None:
                va = ops.cross_input(video_output, audio_output, P["video_frame_embedding"], P["video_type_embeddings"],
                                     P["audio_frame_embedding"], P["audio_type_embeddings"])
                ranges = {"tva": (0, Sv + Sa), "tv": (0, Sv), "ta": (Sv, Sa)}
            elif video_output is not None:                        # a task string without audio groups: video rows only
                va = ops.single_input(video_output, P["video_frame_embedding"], P["video_type_embeddings"])
                ranges = {"tv": (0, Sv)}
            else:
                va = ops.single_input(audio_output, P["audio_frame_embedding"], P["audio_type_embeddings"])
                ranges = {"ta": (0, Sa)}
            if defer_kv and self._graph_decoder():
                kv_layers = _DeferredKV(va)          # projected inside the captured decoder stack (_decoder_stack)
            else:
                kv_layers = self.project_cross_kv(va)
        return kv_layers, ranges

    def _graph_decoder(self):
        """the decoder's layer stack + its K|V projections as one graphed segment (training passes, valor_amd/graphs.py): static shapes --
        every token row goes through the layers whatever the masker drew; what depends on the draw (the gather of the masked rows, the
        prediction head, the loss) stays eager behind it. Not with activation checkpointing, not in generation / evaluation."""
        return self._use_graphs() and not self.checkpointing and os.environ.get("VALOR_GRAPH_DECODER", "1") != "0"

    def bert_encoder(self, x, mask, kv_layers, kv_range, kv_bmod, self_attn=None):
        """BertEncoder / BertLayer.forward bert.py:440-518 (post-LN; va_concate cross-attention).
        self_attn (generation with a K|V cache, valor_amd/decode.py): callable (layer, qkv [B, T, 3E]) -> attention output [B, T, E] that
        stands in for the self-attention over the T rows alone."""
        P, H, p = self.P, self.spec.heads, (self.p_drop if self.training else 0.0)
        for i in range(self.spec.layers):
            q = f"multimodal_encoder.encoder.layer.{i}."
            qkv = ops.linear(x, P[q + "attention.self.qkv.weight"], P[q + "attention.self.qkv.bias"])
            a = self_attn(i, qkv) if self_attn is not None else ops.self_attention(qkv, H, mask, p)
            o = ops.linear(a, P[q + "attention.output.dense.weight"], None)
            x = ops.bias_dropout_residual_ln(o, P[q + "attention.output.dense.bias"], x, P[q + "attention.output.LayerNorm.weight"],
                                             P[q + "attention.output.LayerNorm.bias"], 1e-12, p, False)
            if isinstance(kv_layers, _BlockKV):
                x = self._cross_blocks(i, x, kv_layers, kv_range, kv_bmod, p)
            elif kv_layers is not None:
                cq = ops.linear(x, P[q + "cross_attn.cross.query.weight"], P[q + "cross_attn.cross.query.bias"])
                c = ops.cross_attention(cq, kv_layers[i], H, kv_range, kv_bmod, p, grad_slot=self._kv_slots[i])
                o = ops.linear(c, P[q + "cross_attn.output.dense.weight"], None)
                x = ops.bias_dropout_residual_ln(o, P[q + "cross_attn.output.dense.bias"], x, P[q + "cross_attn.output.LayerNorm.weight"],
                                                 P[q + "cross_attn.output.LayerNorm.bias"], 1e-12, p, False)
            m = ops.mlp(x, P[q + "intermediate.dense.weight"], P[q + "intermediate.dense.bias"], P[q + "output.dense.weight"], None, ACT_GELU_ERF)
            x = ops.bias_dropout_residual_ln(m, P[q + "output.dense.bias"], x, P[q + "output.LayerNorm.weight"], P[q + "output.LayerNorm.bias"], 1e-12, p, False)
        return x

    def _cross_blocks(self, i, x, kv, mods, kv_bmod, p):
        """the per-modality cross-attention blocks of layer i (bert.py:459-496): mods = the modalities this query group attends to.
        One modality: its block alone (every mode). Both: 'va_parallel' = both blocks read x and their outputs (each a full BertAttention:
        attention, dense, dropout, residual, LayerNorm) are summed; 'video_audio' / 'audio_video' = one after the other."""
        P, H = self.P, self.spec.heads

        def block(m, xin):
            q = f"multimodal_encoder.encoder.layer.{i}.cross_attn_{m}."
            cq = ops.linear(xin, P[q + "cross.query.weight"], P[q + "cross.query.bias"])
            c = ops.cross_attention(cq, kv.of(m)[i], H, None, kv_bmod, p)
            o = ops.linear(c, P[q + "output.dense.weight"], None)
            return ops.bias_dropout_residual_ln(o, P[q + "output.dense.bias"], xin, P[q + "output.LayerNorm.weight"],
                                                P[q + "output.LayerNorm.bias"], 1e-12, p, False)
        if len(mods) == 1:
            return block(mods[0], x)
        if self.spec.cross_attn_type == "va_parallel":
            return block("v", x) + block("a", x)
        for m in (("v", "a") if self.spec.cross_attn_type == "video_audio" else ("a", "v")):
            x = block(m, x)
        return x

    def cls_transform(self, rows):
        """BERTPredictionHead dense -> GELU -> LayerNorm (modeling.py:249-252); the decoder GEMM follows."""
        P = self.P
        h = ops.linear(rows, P["cls.dense.weight"], P["cls.dense.bias"], ACT_GELU_ERF)
        return ops.layer_norm(h, P["cls.layernorm.weight"], P["cls.layernorm.bias"], 1e-12)

    def _decoder_groups(self, txt_input, txt_labels, groups, prompt_cpu, casual, kv_layers, ranges, b, compute_loss, tag, out, per_sample=False,
                        kv_b=None):
        """Run the decoder for len(groups) query groups as ONE batch (same text input, different K/V rows)."""
        blocks = isinstance(kv_layers, _BlockKV)
        if blocks and len(groups) > 1:
            # a block per modality: the groups differ in which blocks a layer runs (bert.py:459-496), so they cannot share a batch
            res = [self._decoder_groups(txt_input, txt_labels, [g], prompt_cpu, casual, kv_layers, ranges, b, compute_loss, tag, out, per_sample, kv_b)
                   for g in groups]
            if compute_loss and per_sample:
                return torch.cat([r[0] for r in res], dim=0), res[0][1]
            return sum(res) / len(res) if compute_loss else None        # equal row counts per group: the mean of the per-group means
        G, T = len(groups), txt_input.shape[1]
        ids = self._dev(txt_input)
        x = self._bert_embed(ids, T, None, self._full_attn and casual)
        if prompt_cpu is not None:
            xp = self._bert_embed(self._dev(prompt_cpu), prompt_cpu.shape[1], "prompt")
            x = torch.cat((x, xp), dim=1)
        Ttot = x.shape[1]
        mask = self._dev(self._bert_mask(txt_input, prompt_cpu, casual, self._full_attn and casual))
        if G > 1:
            x = x.repeat(G, 1, 1)
            mask = mask.repeat(G, 1, 1)
        kv_range = None
        if blocks:
            kv_range = ranges[groups[0]]
        elif kv_layers is not None:
            kv_range = self._dev(torch.tensor([list(ranges[g]) for g in groups for _ in range(b)], dtype=torch.int32))
        # kv_b: the K|V batch when it is smaller than the text rows (row r attends to clip r % kv_b: answer-major tiled rows of image QA)
        hidden = self.bert_encoder(x, mask, kv_layers, kv_range, (kv_b or b) if kv_layers is not None else 0)
        sel = (txt_labels != -1)
        bi, tj = sel.nonzero(as_tuple=True)                      # host tensors, row-major order == boolean indexing order
        n = bi.numel()
        idx = self._dev(torch.cat([(g * b + bi) * Ttot + tj for g in range(G)]))
        rows = ops.gather_rows(hidden.reshape(-1, hidden.shape[-1]), idx)
        h = self.cls_transform(rows)
        P = self.P
        labels = self._dev(txt_labels[sel].repeat(G))
        if compute_loss and per_sample:
            # forward_qa_single (pretrain.py:1282-1290): CE summed per sample / that sample's masked-token count, then the mean over
            # samples (and over groups: G * b equally weighted segments). Rows are ordered (group, sample, position).
            counts = sel.sum(dim=1).tolist()
            rows = [r for r, c in enumerate(counts) if c > 0]            # padding rows of a tiled batch carry no labels
            losses = ops.decoder_xent_segments(h, P["multimodal_encoder.embeddings.word_embeddings.weight"], P["cls.decoder.bias"], labels,
                                               [counts[r] for r in rows] * G)
            return torch.stack(losses).view(G, len(rows)), rows
        if compute_loss:
            # equal row counts per group: the mean over all G*n rows == mean of the per-group means (pretrain.py:473-479)
            return ops.decoder_xent(h, P["multimodal_encoder.embeddings.word_embeddings.weight"], P["cls.decoder.bias"], labels,
                                    smoothing=self._smoothing)
        scores = ops.decoder_logits(h, P["multimodal_encoder.embeddings.word_embeddings.weight"], P["cls.decoder.bias"])
        for gi, g in enumerate(groups):
            out[f"{tag}_scores_{g}"] = scores[gi * n:(gi + 1) * n]
        return None

    def _decoder_layer(self, i, X, kv, ssegs, xsegs, dkv):
        """one BertLayer (bert.py:440-496) on the row-batched stack of every decoder pass"""
        P, H = self.P, self.spec.heads
        p = self.p_drop if self.training else 0.0
        q = f"multimodal_encoder.encoder.layer.{i}."
        # post-LN: every sub-layer input feeds the sub-layer's first GEMM AND the residual add behind it; the two gradients meet in a
        # GradSlot (the LayerNorm backward publishes its residual gradient, the GEMM's dgrad accumulates into it: no add kernels)
        s1, s2, s3 = ops.GradSlot(), ops.GradSlot(), ops.GradSlot()
        qkv = ops.linear(X, P[q + "attention.self.qkv.weight"], P[q + "attention.self.qkv.bias"], grad_slot=s1)
        a = ops.seg_self_attention(qkv, H, ssegs, p)
        o = ops.linear(a, P[q + "attention.output.dense.weight"], None)
        X = ops.bias_dropout_residual_ln(o, P[q + "attention.output.dense.bias"], X, P[q + "attention.output.LayerNorm.weight"],
                                         P[q + "attention.output.LayerNorm.bias"], 1e-12, p, False, res_slot=s1)
        if kv is not None:
            cq = ops.linear(X, P[q + "cross_attn.cross.query.weight"], P[q + "cross_attn.cross.query.bias"], grad_slot=s2)
            c = ops.seg_cross_attention(cq, kv, H, xsegs, p, dkv_buf=dkv[i] if dkv else None)
            o = ops.linear(c, P[q + "cross_attn.output.dense.weight"], None)
            X = ops.bias_dropout_residual_ln(o, P[q + "cross_attn.output.dense.bias"], X, P[q + "cross_attn.output.LayerNorm.weight"],
                                             P[q + "cross_attn.output.LayerNorm.bias"], 1e-12, p, False, res_slot=s2)
        m = ops.mlp(X, P[q + "intermediate.dense.weight"], P[q + "intermediate.dense.bias"], P[q + "output.dense.weight"], None, ACT_GELU_ERF,
                    grad_slot=s3)
        return ops.bias_dropout_residual_ln(m, P[q + "output.dense.bias"], X, P[q + "output.LayerNorm.weight"], P[q + "output.LayerNorm.bias"], 1e-12, p,
                                            False, res_slot=s3)

    def _decoder_stack(self, X, va, *rest):
        """what the graphed decoder segment runs (and captures): the K|V projections of every layer on the side stream -- a parallel branch
        of the graph, layer i waits for its own projection -- and the layers on the row stack. rest = the passes' attention masks, then
        their key ranges; the passes' geometry comes from self._dec_meta (part of the capture key)."""
        smeta, xmeta = self._dec_meta
        masks, kvrs = rest[:len(smeta)], rest[len(smeta):]
        ssegs = [(r0, Bp, Tt, m) for (r0, Bp, Tt), m in zip(smeta, masks)]
        xsegs = [(r0, Bp, Tt, k, b) for (r0, Bp, Tt, b), k in zip(xmeta, kvrs)]
        kv_layers = self.project_cross_kv(va)
        dkv = getattr(self, "_dkv_static", None)
        for i in range(self.spec.layers):
            X = self._decoder_layer(i, X, kv_layers[i], ssegs, xsegs, dkv)
        return X

    def _decoder_fused(self, passes, kv_layers, ranges, b):
        """Training path: ALL decoder passes (caption groups, every mlm group) as one row-batched stack -- every GEMM /
        LayerNorm of a BertLayer (bert.py:440-496) runs once on the concatenated rows; self- and cross-attention run per
        pass on row segments. passes = [(tag, txt_input, txt_labels, groups, prompt_cpu, casual)]. Returns {tag: [loss]}."""
        P, H, E = self.P, self.spec.heads, self.spec.hidden
        p = self.p_drop if self.training else 0.0
        xs, ssegs, xsegs, idxs, labs, seg_rows, r0 = [], [], [], [], [], [], 0
        for (tag, txt_input, txt_labels, groups, prompt_cpu, casual) in passes:
            G, T = len(groups), txt_input.shape[1]
            x = self._bert_embed(self._dev(txt_input), T, None, self._full_attn and casual)
            if prompt_cpu is not None:
                x = torch.cat((x, self._bert_embed(self._dev(prompt_cpu), prompt_cpu.shape[1], "prompt")), dim=1)
            Ttot = x.shape[1]
            mask = self._dev(self._bert_mask(txt_input, prompt_cpu, casual, self._full_attn and casual))
            if G > 1:
                x = x.repeat(G, 1, 1)
                mask = mask.repeat(G, 1, 1)
            Bp = G * b
            ssegs.append((r0, Bp, Ttot, mask))
            if kv_layers is not None:
                kvr = self._dev(torch.tensor([list(ranges[g]) for g in groups for _ in range(b)], dtype=torch.int32))
                xsegs.append((r0, Bp, Ttot, kvr, b))
            sel = (txt_labels != -1)
            bi, tj = sel.nonzero(as_tuple=True)
            idxs.append(torch.cat([r0 + (g * b + bi) * Ttot + tj for g in range(G)]))
            labs.append(txt_labels[sel].repeat(G))
            seg_rows.append(G * bi.numel())
            xs.append(x.reshape(-1, E))
            r0 += Bp * Ttot
        # the bigger pass first: in backward it writes the shared dK|dV buffer, the others accumulate into it
        X = torch.cat(xs, dim=0) if len(xs) > 1 else xs[0]
        if isinstance(kv_layers, _DeferredKV):
            # the K|V projections + the twelve layers as ONE graphed segment (valor_amd/graphs.py): inputs = the token rows and the
            # [video | audio] rows (both differentiable), the passes' attention masks and key ranges; their geometry is the capture key
            seg = self._graph_segs.get("decoder")
            if seg is None:
                from .. import graphs
                seg = self._graph_segs["decoder"] = graphs.GraphedSegment("decoder", self._decoder_stack)
            self._dec_meta = ([(r0_, Bp_, Tt_) for (r0_, Bp_, Tt_, _m) in ssegs], [(r0_, Bp_, Tt_, b_) for (r0_, Bp_, Tt_, _k, b_) in xsegs])
            X = seg(X, kv_layers.va, *[m for (_a, _b, _c, m) in ssegs], *[k for (_a, _b, _c, k, _d) in xsegs],
                    key_extra=(tuple(self._dec_meta[0]), tuple(self._dec_meta[1])))
        else:
            dkv = getattr(self, "_dkv_static", None)
            # bert.py:510-513: with `checkpointing` every BertLayer keeps its input rows (and the layer's projected K|V, which lives in a
            # static buffer anyway) and runs again in backward; the GradSlots of a layer are rebuilt by that second run
            ckpt = self.checkpointing and self.training and torch.is_grad_enabled()
            for i in range(self.spec.layers):
                kv = kv_layers[i] if kv_layers is not None else None
                if not ckpt:
                    X = self._decoder_layer(i, X, kv, ssegs, xsegs, dkv)
                elif kv is None:
                    X = ops.checkpoint(lambda X_, i=i: self._decoder_layer(i, X_, None, ssegs, xsegs, dkv), X)
                else:
                    X = ops.checkpoint(lambda X_, kv_, i=i: self._decoder_layer(i, X_, kv_, ssegs, xsegs, dkv), X, kv)
        rows = ops.gather_rows(X, self._dev(torch.cat(idxs)))
        h = self.cls_transform(rows)
        losses = ops.decoder_xent_segments(h, P["multimodal_encoder.embeddings.word_embeddings.weight"], P["cls.decoder.bias"],
                                           self._dev(torch.cat(labs)), seg_rows, smoothing=self._smoothing)
        res = {}
        for (tag, *_), l in zip(passes, losses):
            res.setdefault(tag, []).append(l)
        return res

    # ------------------------------------------------------------------ BASELINE configs[0]
    def text_mlm(self, batch, compute_loss=True):
        """Text-only MLM step (the reference's CPU-runnable plumbing case; forward_pt cannot express it):
        TokenMasker p = 0.15 -> multimodal encoder without cross-attention input, casual=False -> prediction head on the masked
        rows -> cross-entropy. Same kernels as the decoder passes of forward_pt."""
        self.stage.begin_step()
        txt = batch["txt_tokens"]["bert_tokens"].cpu()
        mlm_in, mlm_lab = self.text_masker(txt, 0.15)
        out = {}
        loss = self._decoder_groups(mlm_in, mlm_lab, ["t"], None, False, None, {}, txt.shape[0], compute_loss, "mlm", out)
        if compute_loss:
            return {"mlm_loss": loss}
        out["txt_labels_mlm"] = mlm_lab
        return out

    # ------------------------------------------------------------------ the hot path
    def forward(self, batch, task, compute_loss=True):
        """VALOR.forward, model/pretrain.py:125-135"""
        if task.startswith("pt"):
            return self.forward_pt(batch, task, compute_loss=compute_loss)
        if task.startswith("ret"):
            return self.forward_ret(batch, task, compute_loss=compute_loss)
        if task.startswith("cap"):
            return self.forward_cap(batch, task, compute_loss=compute_loss)
        if task.startswith("qa"):
            return self.forward_qa(batch, task, compute_loss=compute_loss)
        raise NotImplementedError(f"task {task!r}: 'pt_*', 'ret%*', 'cap%*' and 'qa%*' are the reference's task families (pretrain.py:125-135)")

    def forward_ret(self, batch, task, compute_loss=True):
        """VALOR.forward_ret, model/pretrain.py:544-711 (config/fast-retrieval-*.json: 'ret%tva%tv'): the contrastive branch of forward_pt
        (:252-407 -- the same encoders, pooling, heads, gathers, fine matrices and InfoNCE, line for line) on the groups after 'ret%';
        the mean of the group losses is NOT scaled by contra_loss_ratio (:706 vs :406). compute_loss=False returns
        feat_t / feat_v / feat_a / txt_tokens (:708-715), what evaluate.compute_fine_matrix consumes."""
        return self._forward_groups(batch, [], [], task.split("%")[1:], compute_loss, contra_ratio=1.0)

    def forward_cap(self, batch, task, compute_loss=True):
        """VALOR.forward_cap, model/pretrain.py:713-725 (config/caption-*.json: 'cap%tva%tv'; caption_type 'unimlm', no label smoothing /
        scst / full_masker -- the shipped settings). Loss: forward_cap_single :802-880 = the caption passes of forward_pt. Otherwise
        generate_cap :914-985 -> valor_amd.decode (greedy for beam_size 1, beam search above)."""
        groups = task.split("%")[1:]
        if compute_loss:
            self._smoothing, self._full_attn = self.label_smoothing, self.full_masker      # pretrain.py:835-860: forward_cap_single passes full_masker
            try:
                return self._forward_groups(batch, [], groups, [], True)
            finally:
                self._smoothing, self._full_attn = 0.0, False
        from .. import decode
        return decode.generate_cap(self, batch, groups)

    def qa_prompt(self, question_cpu):
        """the prompt rows of the QA passes are the QUESTION (prompt-type embeddings), 'answer the question' spliced in behind its [CLS]
        when use_task_prompt (model/pretrain.py:1268-1274)"""
        if not self.use_task_prompt:
            return question_cpu
        tp = self.get_task_prompt(PROMPTS["qa"], question_cpu.shape[0])[:, 1:-1]
        return torch.cat((question_cpu[:, 0:1], tp, question_cpu[:, 1:]), dim=1)

    def forward_qa(self, batch, task, compute_loss=True):
        """VALOR.forward_qa, model/pretrain.py:1191-1459 (config/VQA-*.json: 'qa%tva%tv'). Loss: forward_qa_single :1212-1345 --
        TokenMasker p = 0.99 on the answer rows, causal decoder with the question as prompt rows, per-row-normalised CE; one answer per
        question (video QA: the mean over samples) or several weighted candidates (image QA, `answer_nums` / `answer_weights`: weighted
        rows summed over the question count). Otherwise generate_qa :1366-1459 -> valor_amd.decode with beam_size_qa (one question per
        clip)."""
        groups = [g for g in ("tva", "tv", "ta") if g in task.split("%")[1:]]
        prompt = self.qa_prompt(batch["question_tokens"]["bert_tokens"].cpu())
        if not compute_loss:
            from .. import decode
            return decode.generate_qa(self, batch, groups, prompt)
        self.stage.begin_step()
        txt = batch["txt_tokens"]["bert_tokens"].cpu()
        nums = [int(n) for n in batch.get("answer_nums", [1] * txt.shape[0])]
        b = len(nums)
        qa_in, qa_lab = self.caption_inputs(txt, 0.99)                 # in the reference's row order (sample-major): the draw order is the contract
        weights = None
        if any(n != 1 for n in nums):
            # image QA (pretrain.py:1243-1265): the reference tiles question / video / audio rows per candidate answer. Here the answer
            # rows are laid out ANSWER-major and padded to max(nums) * b rows, so that row r belongs to clip r % b -- the K|V addressing
            # the cross-attention kernels have -- and no K|V row is copied; padding rows repeat a real row with every label -1.
            nmax, start = max(nums), np.concatenate(([0], np.cumsum(nums)))
            src = np.array([start[i] + (j if j < nums[i] else 0) for j in range(nmax) for i in range(b)])
            valid = np.array([j < nums[i] for j in range(nmax) for i in range(b)])
            qa_in, qa_lab = qa_in[src], qa_lab[src].clone()
            qa_lab[torch.from_numpy(~valid)] = -1
            prompt = prompt.repeat(nmax, 1)
            w = torch.as_tensor(batch["answer_weights"], dtype=torch.float32).reshape(-1)
            weights = w[src[valid]]                                  # rows with labels, in the padded order
        alltasks = "".join(groups)
        video_output = self.forward_video_encoder(batch["video_pixels"]) if "v" in alltasks else None
        audio_output = self.forward_audio_encoder(batch["audio_spectrograms"]) if "a" in alltasks else None
        kv_layers, ranges = self.cross_inputs(video_output, audio_output)
        self._full_attn = self.full_masker                           # pretrain.py:1276,1300,1324: forward_qa_single passes full_masker
        try:
            L, rows = self._decoder_groups(qa_in, qa_lab, groups, prompt, True, kv_layers, ranges, qa_in.shape[0], True, "qa", {}, per_sample=True, kv_b=b)
        finally:
            self._full_attn = False
        if weights is None:
            return {"qa_loss": L.mean()}                            # mean over samples, mean over groups (:1290,1338-1343)
        return {"qa_loss": ((L * self._dev(weights)).sum(dim=1) / b).mean()}      # weighted rows summed over the QUESTION count (:1288-1289)

    def forward_pt(self, batch, task, compute_loss=True):
        """VALOR.forward_pt, model/pretrain.py:214-541."""
        mlm_task, caption_task, contra_task = [], [], []
        for i in task.split("_"):
            if "mlm" in i:
                mlm_task = i.split("%")[1:]
            elif "caption" in i:
                caption_task = i.split("%")[1:]
            elif "contra" in i:
                contra_task = i.split("%")[1:]
        return self._forward_groups(batch, mlm_task, caption_task, contra_task, compute_loss, contra_ratio=self.contra_loss_ratio)

    def caption_inputs(self, txt, mask_prob=0.6):
        """inputs / labels of the caption passes (model/pretrain.py:424-433, :807-816; the answer rows of QA at 0.99, :1225-1234): caption_type 'unimlm' = TokenMasker; 'lm' =
        the tokens as they are, label = the NEXT token (0 = padding and the last position: ignored, -1)"""
        if self.caption_type == "unimlm" and self.full_masker:          # full_mask, pretrain.py:137-142
            n = txt.shape[1]
            tokens = torch.cat((txt, torch.full_like(txt, self.text_mask_token)), dim=1)
            labels = -torch.ones_like(tokens)
            nz = txt[:, 1:n] != 0
            labels[:, n:2 * n - 1][nz] = txt[:, 1:n][nz]
            return tokens, labels
        if self.caption_type == "unimlm":
            return self.text_masker(txt, mask_prob)
        labels = torch.zeros_like(txt)
        labels[:, :txt.shape[1] - 1] = txt[:, 1:]
        labels[labels == 0] = -1
        return txt, labels

    def _forward_groups(self, batch, mlm_task, caption_task, contra_task, compute_loss, contra_ratio=1.0):
        """The body of VALOR.forward_pt (model/pretrain.py:226-541) on parsed group lists; forward_ret / forward_cap run it with one branch."""
        P, sp = self.P, self.spec
        self.stage.begin_step()
        if self.device.type == "cuda":
            streams.set_main(self.device)
            if torch.is_grad_enabled():
                ops.K.ReduceQueue.discard_stale()    # leftovers of a backward pass that died half way (kernels.ReduceQueue)
            if self._use_graphs():
                # the graphed segments key their captures on the by-value dropout offset at their entry, which only repeats if the step's
                # driver restarts it (TrainEngine.train_step does): a custom loop that never calls begin_step() gets it from here
                if not ops.DropoutState.begun:
                    ops.DropoutState.begin_step()
                ops.DropoutState.begun = False
        out = {}
        col = self.collect
        txt_tokens = batch.get("txt_tokens")
        # Host-side token masking FIRST, in the reference's order (the masker consumes the python RNG: caption
        # pretrain.py:428, then mlm :488; nothing else on this path does). Done before any kernel of this step is queued,
        # its Python loops overlap the GPU's tail of the previous step instead of draining the pipeline mid-forward.
        cap_in = cap_lab = mlm_in = mlm_lab = None
        if caption_task and self.full_masker and not self._full_attn:
            # forward_pt builds the doubled rows (pretrain.py:425-426) but slices the 'tv' / 'ta' outputs with the ORIGINAL length against the
            # doubled labels (:454, :466): an IndexError in the reference. Only the finetune paths (forward_cap / forward_qa) work with it.
            # (A caption task of 'tva' ONLY does run in the reference -- its tv / ta slices are what fails; that one case is refused here too.)
            raise NotImplementedError("full_masker with a pretraining caption task fails in the reference too (model/pretrain.py:454, the tv / ta "
                                      "slices; a 'tva'-only caption task is the one case the reference runs and this model refuses); "
                                      "use it with 'cap%..' / 'qa%..'")
        if caption_task or mlm_task:
            txt = txt_tokens["bert_tokens"].cpu()
            if caption_task:
                cap_in, cap_lab = self.caption_inputs(txt)
            if mlm_task:
                mlm_in, mlm_lab = self.text_masker(txt, 0.15)
        alltasks = "".join(mlm_task + caption_task + contra_task)
        video_output = audio_output = txt_output = None
        clip_text = "t" in "".join(contra_task) and sp.txt_encoder != "bert"
        # the audio encoder and the CLIP text tower on a second stream beside the video encoder (valor_amd/streams.py); the shared-BERT
        # text pass stays on this stream (its weight gradients land in the same arena slots as the decoder's)
        side = None
        if streams.enabled() and self.device.type == "cuda" and "v" in alltasks and ("a" in alltasks or clip_text):
            side = streams.side_stream(self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if "a" in alltasks:
                    audio_output = self.forward_audio_encoder(batch["audio_spectrograms"])
                if clip_text:
                    clip_tokens = txt_tokens["clip_tokens"].cpu()
                    txt_output = self.forward_txt_encoder(clip_tokens)
        if "v" in alltasks:
            video_output = self.forward_video_encoder(batch["video_pixels"])
        if side is not None:
            audio_output, txt_output = streams.join(side, audio_output, txt_output)
        if "a" in alltasks and audio_output is None:
            audio_output = self.forward_audio_encoder(batch["audio_spectrograms"])
        if "t" in "".join(contra_task) and txt_output is None:
            if sp.txt_encoder == "bert":                          # get_text_tokens: the bert ids (pretrain.py:252)
                clip_tokens = txt_tokens["bert_tokens"].cpu()
                txt_output = self.forward_txt_encoder_bert(clip_tokens)
            else:
                clip_tokens = txt_tokens["clip_tokens"].cpu()
                txt_output = self.forward_txt_encoder(clip_tokens)
        if col is not None:
            col.update(video_output=video_output, audio_output=audio_output, txt_output=txt_output)

        # ---------------- contra_type 'coarse' (pretrain.py:266-291,375-395; pooling modeling.py:373-407): one pooled vector per modality
        if contra_task and sp.contra_type == "coarse":
            feat_t = feat_v = feat_a = None
            if txt_output is not None:
                bt, Lt = txt_output.shape[:2]
                if sp.txt_encoder == "bert":                      # [CLS]
                    rows = torch.arange(bt, dtype=torch.long) * Lt
                    pooled = ops.gather_rows(txt_output.reshape(bt * Lt, -1), self._dev(rows))
                    feat_t = ops.l2_normalize(ops.linear(pooled, P["contra_head_t.linear.weight"], None))
                else:                                              # the EOT row: tokens.argmax (clip.py:425 convention)
                    rows = torch.arange(bt, dtype=torch.long) * Lt + clip_tokens.argmax(dim=-1)
                    pooled = ops.gather_rows(txt_output.reshape(bt * Lt, -1), self._dev(rows))
                    feat_t = ops.l2_normalize(ops.linear(pooled, P["clip_model.text_projection"], None, w_is_kn=True))
            if "v" in "".join(contra_task):
                b, F = video_output.shape[:2]
                if sp.video_encoder == "swin":                     # token mean, then frame mean = the mean over all F * X rows
                    pooled = ops.group_mean(video_output.reshape(-1, sp.video_dim), F * video_output.shape[2])
                else:
                    vo, cls_v = ops.tap_rows(video_output.reshape(-1, sp.vis_width), self._const_idx(b * F, sp.vis_tokens))
                    video_output = vo.view(video_output.shape)          # what the decoder inputs read below (ops.TapRowsFn)
                    pooled = ops.group_mean(cls_v, F)
                if sp.clip_heads:
                    feat_v = ops.l2_normalize(ops.linear(pooled, P["clip_model.visual.proj"], None, w_is_kn=True))
                else:
                    feat_v = ops.l2_normalize(ops.linear(pooled, P["contra_head_v.linear.weight"], None))
            if "a" in "".join(contra_task):
                b, A = audio_output.shape[:2]
                ao, cls_a = ops.tap_rows(audio_output.reshape(-1, sp.aud_width), self._const_idx(b * A, sp.aud_tokens))
                audio_output = ao.view(audio_output.shape)
                feat_a = ops.l2_normalize(ops.linear(ops.group_mean(cls_a, A), P["contra_head_a.linear.weight"], None))
            tok_contra = clip_tokens if txt_output is not None else None
            if compute_loss and self.gather_fn is not None:
                feat_t, feat_v, feat_a, tok_contra = self.gather_fn(feat_t, feat_v, feat_a, self._dev(tok_contra))
            if col is not None:
                col.update(feat_t=feat_t, feat_v=feat_v, feat_a=feat_a)
            if compute_loss:
                k = P["clip_model.logit_scale"].float().exp() if sp.video_encoder == "clip" else 1.0 / P["contra_temp"].float()
                losses = []
                for g in contra_task:
                    if g == "tv":
                        losses.append(ops.coarse_contrastive(feat_t, [feat_v], k))
                    elif g == "ta":
                        losses.append(ops.coarse_contrastive(feat_t, [feat_a], k))
                    elif g == "tva" and sp.late_fusion:            # the tv and ta matrices summed (:383-384)
                        losses.append(ops.coarse_contrastive(feat_t, [feat_v, feat_a], k))
                    elif g == "tva":                               # va_fusion(cat(v, a)) (:385-387)
                        fva = ops.l2_normalize(ops.linear(torch.cat((feat_v, feat_a), dim=-1), P["va_fusion.weight"], P["va_fusion.bias"]))
                        losses.append(ops.coarse_contrastive(feat_t, [fva], k))
                    else:
                        raise NotImplementedError(f"contrastive group {g} with contra_type='coarse' (the reference handles tv / tva / ta, pretrain.py:375-395)")
                out["contra_loss"] = sum(losses) / len(losses) * contra_ratio
            else:
                out.update(feat_t=feat_t, feat_v=feat_v, feat_a=feat_a, txt_tokens=tok_contra)
        # ---------------- MGA contrastive (pretrain.py:266-407)
        elif contra_task:
            feat_t = feat_v = feat_a = None
            tok_contra = None
            if txt_output is not None:
                if sp.txt_encoder == "bert":                      # Contra_head (pretrain.py:33-38,94)
                    feat_t = ops.l2_normalize(ops.linear(txt_output, P["contra_head_t.linear.weight"], None))
                else:
                    feat_t = ops.l2_normalize(ops.linear(txt_output, P["clip_model.text_projection"], None, w_is_kn=True))
                tok_contra = clip_tokens
            if "v" in "".join(contra_task) and sp.video_encoder == "swin":
                b, F, X = video_output.shape[:3]                  # mean over the frame's tokens (modeling.py:388-389)
                pooled = ops.group_mean(video_output.reshape(-1, sp.video_dim), X)
                feat_v = ops.l2_normalize(ops.linear(pooled, P["contra_head_v.linear.weight"], None)).view(b, F, -1)
            elif "v" in "".join(contra_task):
                b, F = video_output.shape[:2]
                idx = self._const_idx(b * F, sp.vis_tokens)
                vo, cls_v = ops.tap_rows(video_output.reshape(-1, sp.vis_width), idx)
                video_output = vo.view(video_output.shape)              # what the decoder inputs read below (ops.TapRowsFn)
                if sp.clip_heads:                                  # pretrain.py:89-92
                    feat_v = ops.l2_normalize(ops.linear(cls_v, P["clip_model.visual.proj"], None, w_is_kn=True)).view(b, F, -1)
                else:                                              # Contra_head beside a CLIP video encoder (pretrain.py:93-97)
                    feat_v = ops.l2_normalize(ops.linear(cls_v, P["contra_head_v.linear.weight"], None)).view(b, F, -1)
            if "a" in "".join(contra_task):
                b, A = audio_output.shape[:2]
                idx = self._const_idx(b * A, sp.aud_tokens)
                ao, cls_a = ops.tap_rows(audio_output.reshape(-1, sp.aud_width), idx)
                audio_output = ao.view(audio_output.shape)
                feat_a = ops.l2_normalize(ops.linear(cls_a, P["contra_head_a.linear.weight"], None)).view(b, A, -1)
            if compute_loss and self.gather_fn is not None:       # ddp_allgather_with_grads / ddp_allgather (pretrain.py:278-291)
                feat_t, feat_v, feat_a, tok_contra = self.gather_fn(feat_t, feat_v, feat_a, self._dev(tok_contra))
            if col is not None:
                col.update(feat_t=feat_t, feat_v=feat_v, feat_a=feat_a)
            if compute_loss:
                if sp.video_encoder == "clip":
                    k = P["clip_model.logit_scale"].float().exp()                # 1/temp, modeling.py:420-426
                else:
                    k = 1.0 / P["contra_temp"].float()
                maskA = None
                if tok_contra is not None:
                    maskA = (tok_contra != 0).float().contiguous() if tok_contra.is_cuda else self._dev((tok_contra != 0).float()).contiguous()
                fw = lambda name, f: ops.rowdot(ops.linear(f, P[f"{name}_fine_weight.0.weight"], P[f"{name}_fine_weight.0.bias"], ACT_RELU),
                                                P[f"{name}_fine_weight.2.weight"], P[f"{name}_fine_weight.2.bias"]).float().squeeze(-1)
                wt = fw("text", feat_t) if feat_t is not None else None
                wv = fw("video", feat_v) if feat_v is not None else None
                wa = fw("audio", feat_a) if feat_a is not None else None
                ones = lambda f: torch.ones(f.shape[:2], dtype=torch.float32, device=self.device)
                losses = []
                if "tva" in contra_task and sp.late_fusion:       # pretrain.py:313-321: fine(t, v) + fine(t, a) with unit token weights
                    losses.append(ops.late_fusion_fine_contrastive(feat_t, feat_v, feat_a, maskA, k))
                elif "tva" in contra_task:
                    fB, wB = torch.cat((feat_v, feat_a), dim=1), torch.cat((wv, wa), dim=1)
                    losses.append(ops.fine_contrastive(feat_t, fB, wt.contiguous(), wB.contiguous(), maskA, ones(fB), k))
                if "tv" in contra_task:
                    losses.append(ops.fine_contrastive(feat_t, feat_v, wt.contiguous(), wv.contiguous(), maskA, ones(feat_v), k))
                if "ta" in contra_task:
                    losses.append(ops.fine_contrastive(feat_t, feat_a, wt.contiguous(), wa.contiguous(), maskA, ones(feat_a), k))
                if "va" in contra_task:                            # pretrain.py:346-352: video tokens against audio tokens
                    losses.append(ops.fine_contrastive(feat_v, feat_a, wv.contiguous(), wa.contiguous(), ones(feat_v), ones(feat_a), k))
                if "vta" in contra_task:                           # :354-361: video queries against [text | audio]
                    fB, wB = torch.cat((feat_t, feat_a), dim=1), torch.cat((wt, wa), dim=1)
                    losses.append(ops.fine_contrastive(feat_v, fB, wv.contiguous(), wB.contiguous(), ones(feat_v),
                                                       torch.cat((maskA, ones(feat_a)), dim=1).contiguous(), k))
                if "atv" in contra_task:                           # :363-370: audio queries against [text | video]
                    fB, wB = torch.cat((feat_t, feat_v), dim=1), torch.cat((wt, wv), dim=1)
                    losses.append(ops.fine_contrastive(feat_a, fB, wa.contiguous(), wB.contiguous(), ones(feat_a),
                                                       torch.cat((maskA, ones(feat_v)), dim=1).contiguous(), k))
                for g in contra_task:
                    if g not in ("tva", "tv", "ta", "va", "vta", "atv"):
                        raise NotImplementedError(f"contrastive group {g}")
                out["contra_loss"] = sum(losses) / len(losses) * contra_ratio
            else:
                out.update(feat_t=feat_t, feat_v=feat_v, feat_a=feat_a, txt_tokens=tok_contra)

        # ---------------- decoder inputs (pretrain.py:410-416, modeling.py:485-502)
        if not (caption_task or mlm_task):
            return out
        txt = txt_tokens["bert_tokens"].cpu()
        bs = txt.shape[0]
        kv_layers, ranges = self.cross_inputs(video_output, audio_output, defer_kv=compute_loss)

        if compute_loss:
            # training: every decoder pass row-batched into one stack (caption groups first: the biggest segment)
            passes = []
            if caption_task:
                groups = [g for g in ("tva", "tv", "ta") if g in caption_task]
                prompt = self.get_task_prompt(PROMPTS["caption"], bs) if self.use_task_prompt else None
                passes.append(("caption", cap_in, cap_lab, groups, prompt, True))
            for g in ("tva", "tv", "ta"):
                if g in mlm_task:
                    passes.append(("mlm", mlm_in, mlm_lab, [g], self.get_task_prompt(PROMPTS["mlm_" + g], bs), False))
            if isinstance(kv_layers, _BlockKV):          # a cross-attention block per modality: one decoder run per pass and group
                res = {}
                for (tag, tin, tlab, groups, prompt, casual) in passes:
                    res.setdefault(tag, []).append(self._decoder_groups(tin, tlab, groups, prompt, casual, kv_layers, ranges, bs, True, tag, out))
            else:
                res = self._decoder_fused(passes, kv_layers, ranges, bs)
            if "caption" in res:
                out["caption_loss"] = res["caption"][0]                                  # pretrain.py:473-479
            if "mlm" in res:
                out["mlm_loss"] = sum(res["mlm"]) / len(res["mlm"])                      # pretrain.py:524-532
            return out
        # evaluation (compute_loss=False, pretrain.py:445-446,499-500): per-pass scores
        if mlm_task:                                                              # pretrain.py:483-535
            for g in ("tva", "tv", "ta"):
                if g in mlm_task:
                    prompt = self.get_task_prompt(PROMPTS["mlm_" + g], bs)
                    self._decoder_groups(mlm_in, mlm_lab, [g], prompt, False, kv_layers, ranges, bs, False, "mlm", out)
            out["txt_labels_mlm"] = mlm_lab
        if caption_task:                                                          # pretrain.py:419-481
            groups = [g for g in ("tva", "tv", "ta") if g in caption_task]
            prompt = self.get_task_prompt(PROMPTS["caption"], bs) if self.use_task_prompt else None
            self._decoder_groups(cap_in, cap_lab, groups, prompt, True, kv_layers, ranges, bs, False, "caption", out)
            out["txt_labels_caption"] = cap_lab
        return out
