"""TEST INFRASTRUCTURE -- CPU oracle: a plain-PyTorch fp32 restatement of the reference VALOR
pretraining step (CLIP-ViT variant and VideoSwin + BERT-text variant). NOT product code: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it. The product (valor_amd) never does.

Every function cites the reference file:line it follows (/root/reference). The restatement is
pinned against the unmodified reference in this container by tests/test_oracle_vs_reference.py
(runs only where /root/reference exists) and against the committed golden vectors
tests/golden/*.pt (generated from the REFERENCE by oracle/make_goldens.py) everywhere.

Weights: a flat dict with the reference's state-dict keys (valor_amd/synth.py layout); tensors may
require grad so torch autograd provides the reference gradients.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- small pieces
def gelu_erf(x):
    """model/bert.py:52-57, model/transformer.py:32-38"""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def quick_gelu(x):
    """model/clip.py:167-169"""
    return x * torch.sigmoid(1.702 * x)


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def dropout(x, p, training=True):
    return F.dropout(x, p, training) if p > 0 else x


class Oracle:
    def __init__(self, spec, sd, *, dropout_p=0.0, use_task_prompt=False, contra_loss_ratio=1.0, vocab_tokens=None,
                 masker_range=(106, None), drop_path=0.0, caption_type="unimlm", label_smoothing=0.0, full_masker=False):
        assert caption_type in ("unimlm", "lm")
        self.caption_type = caption_type
        self.full_masker = full_masker              # model/pretrain.py:79 (caption_type 'unimlm' only)
        self.label_smoothing = label_smoothing      # model/pretrain.py:72-74: the caption finetune loss only (:839-840)
        self.spec = spec
        self.sd = sd
        self.p = dropout_p
        self.drop_path = drop_path          # VideoSwin stochastic depth rate (videoswin.py:393,418); 0 for parity runs
        self.use_task_prompt = use_task_prompt
        self.contra_loss_ratio = contra_loss_ratio
        self.vocab = {t: i for i, t in enumerate(vocab_tokens)} if vocab_tokens is not None else None
        self.masker_range = (masker_range[0], masker_range[1] if masker_range[1] is not None else spec.vocab)
        self.mask_token = 103

    def w(self, k):
        return self.sd[k]

    # ------------------------------------------------------------------------- CLIP
    def clip_block(self, x, prefix, heads, attn_mask):
        """ResidualAttentionBlock.forward, model/clip.py:186-197 (nn.MultiheadAttention, QuickGELU MLP). x: [N, L, E]"""
        w = self.w
        E = x.shape[-1]
        h = layer_norm(x, w(prefix + "ln_1.weight"), w(prefix + "ln_1.bias"), 1e-5)
        qkv = F.linear(h, w(prefix + "attn.in_proj_weight"), w(prefix + "attn.in_proj_bias"))
        q, k, v = qkv.split(E, dim=-1)
        N, L, _ = q.shape
        hd = E // heads
        q = q.view(N, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
        k = k.view(N, L, heads, hd).transpose(1, 2)
        v = v.view(N, L, heads, hd).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        if attn_mask is not None:
            s = s + attn_mask[:, None]
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(N, L, E)
        x = x + F.linear(a, w(prefix + "attn.out_proj.weight"), w(prefix + "attn.out_proj.bias"))
        h = layer_norm(x, w(prefix + "ln_2.weight"), w(prefix + "ln_2.bias"), 1e-5)
        h = quick_gelu(F.linear(h, w(prefix + "mlp.c_fc.weight"), w(prefix + "mlp.c_fc.bias")))
        return x + F.linear(h, w(prefix + "mlp.c_proj.weight"), w(prefix + "mlp.c_proj.bias"))

    def clip_visual(self, images):
        """VisionTransformer.forward, model/clip.py:259-274: all tokens returned, ln_post on all tokens."""
        w, sp = self.w, self.spec
        x = F.conv2d(images, w("clip_model.visual.conv1.weight"), None, stride=sp.patch)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = w("clip_model.visual.class_embedding") + torch.zeros(x.shape[0], 1, x.shape[-1])
        x = torch.cat([cls, x], dim=1) + w("clip_model.visual.positional_embedding")
        x = layer_norm(x, w("clip_model.visual.ln_pre.weight"), w("clip_model.visual.ln_pre.bias"), 1e-5)
        for i in range(sp.vis_layers):
            x = self.clip_block(x, f"clip_model.visual.transformer.resblocks.{i}.", sp.vis_heads, None)
        return layer_norm(x, w("clip_model.visual.ln_post.weight"), w("clip_model.visual.ln_post.bias"), 1e-5)

    def clip_text(self, tokens):
        """CLIP.encode_text(casual=True), model/clip.py:372-427: causal AND pad additive mask (1-m)*-1e4."""
        w, sp = self.w, self.spec
        x = w("clip_model.token_embedding.weight")[tokens]
        L = x.shape[1]
        x = x + w("clip_model.positional_embedding")[:L]
        m = (tokens != 0).long()
        m = m.unsqueeze(1).expand(-1, L, -1).clone()
        m = torch.tril(m)
        am = (1.0 - m.float()) * -10000.0
        for i in range(sp.txt_layers):
            x = self.clip_block(x, f"clip_model.transformer.resblocks.{i}.", sp.txt_heads, am)
        return layer_norm(x, w("clip_model.ln_final.weight"), w("clip_model.ln_final.bias"), 1e-5)

    # ------------------------------------------------------------------------- VideoSwin
    @staticmethod
    def swin_windows(x, win):
        """window_partition, model/videoswin.py:75-79: [B, D, H, W, C] -> [B*nW, wd*wh*ww, C] (window-major, d/h/w order inside)"""
        B, D, H, W, C = x.shape
        wd, wh, ww = win
        x = x.reshape(B, D // wd, wd, H // wh, wh, W // ww, ww, C)
        return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, wd * wh * ww, C)

    @staticmethod
    def swin_unwindows(xw, win, B, D, H, W):
        """window_reverse, model/videoswin.py:81-84"""
        wd, wh, ww = win
        x = xw.reshape(B, D // wd, H // wh, W // ww, wd, wh, ww, -1)
        return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)

    @staticmethod
    def swin_effective_window(size, window, shift):
        """get_window_size, model/videoswin.py:86-99: a dim no larger than the window gets window = dim, shift = 0"""
        win = tuple(s if s <= w else w for s, w in zip(size, window))
        sh = tuple(0 if s <= w else f for s, w, f in zip(size, window, shift))
        return win, sh

    @staticmethod
    def swin_shift_mask(size, win, shift):
        """compute_mask, model/videoswin.py:272-285 -> [nW, N, N] additive 0 / -100. Region ids per dim: the three slices
        [0, X-w), [X-w, X-s), [X-s, X) written in that order (a zero shift makes the last slice cover, and so relabel, the
        whole dim); two tokens may attend to each other iff all three of their region ids agree."""
        ids = []
        for X, w, s_ in zip(size, win, shift):
            r = torch.zeros(X, dtype=torch.long)
            r[X - w:X - s_ if s_ > 0 else X - w] = 1
            r[X - s_ if s_ > 0 else 0:] = 2
            ids.append(r)
        lab = (ids[0][:, None, None] * 9 + ids[1][None, :, None] * 3 + ids[2][None, None, :]).float()
        lw = Oracle.swin_windows(lab[None, ..., None], win).squeeze(-1)
        diff = lw[:, None, :] - lw[:, :, None]
        return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))

    def swin_attention(self, xw, p, heads, mask):
        """WindowAttention3D.forward, model/videoswin.py:137-163. xw: [B*nW, N, C]"""
        w = self.w
        Bw, N, C = xw.shape
        hd = C // heads
        qkv = F.linear(xw, w(p + "qkv.weight"), w(p + "qkv.bias")).reshape(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        s = q @ k.transpose(-2, -1)
        idx = w(p + "relative_position_index")[:N, :N].reshape(-1)
        bias = w(p + "relative_position_bias_table")[idx].reshape(N, N, heads).permute(2, 0, 1)
        s = s + bias[None]
        if mask is not None:
            nW = mask.shape[0]
            s = (s.view(Bw // nW, nW, heads, N, N) + mask[None, :, None]).view(Bw, heads, N, N)
        a = torch.softmax(s, dim=-1)
        o = (a @ v).transpose(1, 2).reshape(Bw, N, C)
        return F.linear(o, w(p + "proj.weight"), w(p + "proj.bias"))

    def swin_drop_path(self, x, rate):
        """drop_path, model/videoswin.py:40-49 (per-sample stochastic depth, training only)"""
        if rate == 0.0:
            return x
        keep = 1.0 - rate
        m = torch.floor(keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype))
        return x / keep * m

    def swin_block(self, x, p, heads, window, shift, rate):
        """SwinTransformerBlock3D.forward, model/videoswin.py:191-245. x: [B, D, H, W, C]. Feature maps that are not a multiple of
        the window are zero padded AFTER norm1 (:199-203), attend as ordinary slots (their K / V are the qkv bias) and are cropped
        again (:222-223); the shift mask is computed on the padded size (BasicLayer :333-336)."""
        w = self.w
        B, D, H, W, C = x.shape
        win, sh = self.swin_effective_window((D, H, W), window, shift)
        h = layer_norm(x, w(p + "norm1.weight"), w(p + "norm1.bias"), 1e-5)
        pd, pb, pr = (win[0] - D % win[0]) % win[0], (win[1] - H % win[1]) % win[1], (win[2] - W % win[2]) % win[2]
        if pd or pb or pr:
            h = F.pad(h, (0, 0, 0, pr, 0, pb, 0, pd))
        Dp, Hp, Wp = D + pd, H + pb, W + pr
        mask = None
        if any(sh):
            h = torch.roll(h, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
            mask = self.swin_shift_mask((Dp, Hp, Wp), win, sh)
        a = self.swin_attention(self.swin_windows(h, win), p + "attn.", heads, mask)
        a = self.swin_unwindows(a, win, B, Dp, Hp, Wp)
        if any(sh):
            a = torch.roll(a, shifts=sh, dims=(1, 2, 3))
        if pd or pb or pr:
            a = a[:, :D, :H, :W]
        x = x + self.swin_drop_path(a, rate)
        h = layer_norm(x, w(p + "norm2.weight"), w(p + "norm2.bias"), 1e-5)
        h = F.linear(F.gelu(F.linear(h, w(p + "mlp.fc1.weight"), w(p + "mlp.fc1.bias"))), w(p + "mlp.fc2.weight"), w(p + "mlp.fc2.bias"))
        return x + self.swin_drop_path(h, rate)

    def swin_visual(self, video):
        """SwinTransformer3D.forward, model/videoswin.py:441-458; PatchEmbed3D :361-376 (one zero frame appended, conv3d
        kernel (2,4,4) stride (1,4,4), LayerNorm); BasicLayer :329-345 (shift = window // 2 on odd blocks); PatchMerging
        :254-270. video: [b, 3, F, H, W] -> [b, F, H/32 * W/32, C_out]"""
        w, sp = self.w, self.spec
        x = F.pad(video, (0, 0, 0, 0, 0, 1))
        x = F.conv3d(x, w("video_encoder.patch_embed.proj.weight"), w("video_encoder.patch_embed.proj.bias"), stride=(1, 4, 4))
        x = x.permute(0, 2, 3, 4, 1)                                        # [b, D, H/4, W/4, C]
        x = layer_norm(x, w("video_encoder.patch_embed.norm.weight"), w("video_encoder.patch_embed.norm.bias"), 1e-5)
        total = sum(sp.swin_depths)
        rates = [self.drop_path * i / max(total - 1, 1) for i in range(total)]      # torch.linspace(0, rate, total), :418
        shift = tuple(v // 2 for v in sp.swin_window)
        k = 0
        for li, (depth, heads) in enumerate(zip(sp.swin_depths, sp.swin_heads)):
            for bi in range(depth):
                x = self.swin_block(x, f"video_encoder.layers.{li}.blocks.{bi}.", heads, sp.swin_window,
                                    (0, 0, 0) if bi % 2 == 0 else shift, rates[k])
                k += 1
            if li + 1 < len(sp.swin_depths):
                p = f"video_encoder.layers.{li}.downsample."
                if x.shape[2] % 2 or x.shape[3] % 2:                       # PatchMerging pads odd H / W with zeros, :257-259
                    x = F.pad(x, (0, 0, 0, x.shape[3] % 2, 0, x.shape[2] % 2))
                x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], dim=-1)
                x = layer_norm(x, w(p + "norm.weight"), w(p + "norm.bias"), 1e-5)
                x = F.linear(x, w(p + "reduction.weight"))
        x = layer_norm(x, w("video_encoder.norm.weight"), w("video_encoder.norm.bias"), 1e-5)
        return x.reshape(x.shape[0], x.shape[1], -1, x.shape[-1])

    def forward_video_encoder(self, video_pixels):
        """model/modeling.py:449-465"""
        if self.spec.video_encoder == "swin":
            return self.swin_visual(video_pixels.transpose(1, 2))
        b, n, _, h, ww = video_pixels.shape
        out = self.clip_visual(video_pixels.reshape(b * n, 3, h, ww))
        return out.reshape(b, -1, *out.shape[-2:])

    # ------------------------------------------------------------------------- AST
    def forward_audio_encoder(self, audio):
        """model/modeling.py:468-480, AudioEmbeddings :750-762, TransformerEncoder prenorm transformer.py:74-85,156-170"""
        w, sp = self.w, self.spec
        b, n, hh, ww = audio.shape
        x = audio.reshape(-1, hh, ww).unsqueeze(1)
        x = F.conv2d(x, w("audio_embeddings.first_conv.weight"), w("audio_embeddings.first_conv.bias"), stride=sp.aud_patch)
        bb, c = x.shape[0], x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(bb, -1, c)
        x = torch.cat((w("audio_embeddings.cls_token").expand(bb, -1, -1), x), dim=1)
        x = x + w("audio_embeddings.position_embeddings.weight")[None]
        x = dropout(x, self.p)
        H = sp.aud_heads
        for i in range(sp.aud_layers):
            p = f"audio_encoder.layer.{i}."
            res = x
            h = layer_norm(x, w(p + "layernorm1.weight"), w(p + "layernorm1.bias"), 1e-12)
            q, k, v = [F.linear(h, w(p + f"attention.linears.{j}.weight"), w(p + f"attention.linears.{j}.bias"))
                       .view(bb, -1, H, sp.aud_width // H).transpose(1, 2) for j in range(3)]
            att = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(q.shape[-1]), dim=-1)
            att = dropout(att, self.p)
            a = (att @ v).transpose(1, 2).contiguous().view(bb, -1, sp.aud_width)
            a = F.linear(a, w(p + "attention.linears.3.weight"), w(p + "attention.linears.3.bias"))
            x = res + dropout(a, self.p)
            res = x
            h = layer_norm(x, w(p + "layernorm2.weight"), w(p + "layernorm2.bias"), 1e-12)
            h = F.linear(gelu_erf(F.linear(h, w(p + "ff_layer.linear1.weight"), w(p + "ff_layer.linear1.bias"))),
                         w(p + "ff_layer.linear2.weight"), w(p + "ff_layer.linear2.bias"))
            x = res + dropout(h, self.p)
        x = layer_norm(x, w("audio_encoder.last_layernorm.weight"), w("audio_encoder.last_layernorm.bias"), 1e-12)
        return x.reshape(b, n, -1, x.shape[-1])

    # ------------------------------------------------------------------------- BERT decoder
    def bert_embeddings(self, ids, token_type, full_masker=False):
        """BertEmbeddings.forward, model/bert.py:190-218 (full_masker :197-201: the second half of the row -- the [MASK] copies -- sits at
        positions 1 .. L/2, i.e. one past the token each of them has to predict)"""
        w = self.w
        e = "multimodal_encoder.embeddings."
        L = ids.shape[1]
        pos = torch.arange(L)
        if full_masker and token_type is None:
            pos[L // 2:] = pos[:L // 2] + 1
        x = w(e + "word_embeddings.weight")[ids] + w(e + "position_embeddings.weight")[pos][None]
        if token_type == "prompt":
            x = x + w(e + "prompt_embedding.weight")[0]
        else:
            x = x + w(e + "token_type_embeddings.weight")[0]
        x = layer_norm(x, w(e + "LayerNorm.weight"), w(e + "LayerNorm.bias"), 1e-12)
        return dropout(x, self.p)

    def bert_attn(self, x, kv_src, p, blk, mask):
        """BertSelfAttention bert.py:244-289 / BertCrossAttention :314-340 + BertSelfOutput/CrossOutput :351-355,365-371"""
        w, H = self.w, self.spec.heads
        B, T, E = x.shape
        q = F.linear(x, w(p + f"{blk}.query.weight"), w(p + f"{blk}.query.bias"))
        k = F.linear(kv_src, w(p + f"{blk}.key.weight"), w(p + f"{blk}.key.bias"))
        v = F.linear(kv_src, w(p + f"{blk}.value.weight"), w(p + f"{blk}.value.bias"))
        sh = lambda t: t.view(B, -1, H, E // H).permute(0, 2, 1, 3)
        s = sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(E // H)
        if mask is not None:
            s = s + mask
        a = dropout(torch.softmax(s, dim=-1), self.p)
        ctx = (a @ sh(v)).permute(0, 2, 1, 3).contiguous().view(B, T, E)
        out = blk.split(".")[0] + ".output."
        h = dropout(F.linear(ctx, w(p + out + "dense.weight"), w(p + out + "dense.bias")), self.p)
        return layer_norm(h + x, w(p + out + "LayerNorm.weight"), w(p + out + "LayerNorm.bias"), 1e-12)

    def bert_model(self, tokens, task_prompt, video_feat, audio_feat, casual, full_masker=False):
        """BertModel.forward, has_cross_attn branch, model/bert.py:848-896 ; BertLayer :440-496 (every cross_attn_type)"""
        w, sp = self.w, self.spec
        x = self.bert_embeddings(tokens, None, full_masker)
        token_len = x.shape[1]
        am = (tokens != 0).long()
        if task_prompt is not None:
            x = torch.cat((x, self.bert_embeddings(task_prompt, "prompt")), dim=1)
            am = torch.cat((am, (task_prompt != 0).long()), dim=1)
        total = am.shape[1]
        am = am.unsqueeze(1).expand(-1, total, -1).clone()
        if casual and full_masker:                                                               # bert.py:872-878
            n = token_len // 2
            am[:, :n, :n] = torch.tril(am[:, :n, :n])
            am[:, :n, n:token_len] = 0
            am[:, n:token_len, :n] = torch.tril(am[:, n:token_len, :n])
            am[:, n:token_len, n:token_len] = torch.eye(n, dtype=am.dtype)
            am[:, token_len:, :token_len] = 0
        elif casual:
            am[:, :token_len, :token_len] = torch.tril(am[:, :token_len, :token_len])
            am[:, token_len:, :token_len] = 0
        am = ((1.0 - am.unsqueeze(1).float()) * -10000.0)
        mode = getattr(sp, "cross_attn_type", "va_concate")
        if video_feat is not None and audio_feat is not None:
            cross = torch.cat((video_feat, audio_feat), dim=1)
        elif video_feat is not None:
            cross = video_feat
        else:
            cross = audio_feat
        for i in range(sp.layers):
            p = f"multimodal_encoder.encoder.layer.{i}."
            x = self.bert_attn(x, x, p, "attention.self", am)
            if mode == "va_concate":                                                             # bert.py:447-457
                if cross is not None:
                    x = self.bert_attn(x, cross, p, "cross_attn.cross", None)
            elif video_feat is not None and audio_feat is not None:
                if mode == "va_parallel":                                                        # bert.py:459-463: both blocks read x, outputs summed
                    x = self.bert_attn(x, video_feat, p, "cross_attn_v.cross", None) + self.bert_attn(x, audio_feat, p, "cross_attn_a.cross", None)
                else:                                                                            # bert.py:472-476 / 486-489: one after the other
                    first, second = (("v", video_feat), ("a", audio_feat)) if mode == "video_audio" else (("a", audio_feat), ("v", video_feat))
                    x = self.bert_attn(x, first[1], p, f"cross_attn_{first[0]}.cross", None)
                    x = self.bert_attn(x, second[1], p, f"cross_attn_{second[0]}.cross", None)
            elif video_feat is not None:                                                         # one modality: its block alone (every mode)
                x = self.bert_attn(x, video_feat, p, "cross_attn_v.cross", None)
            elif audio_feat is not None:
                x = self.bert_attn(x, audio_feat, p, "cross_attn_a.cross", None)
            h = gelu_erf(F.linear(x, w(p + "intermediate.dense.weight"), w(p + "intermediate.dense.bias")))
            h = dropout(F.linear(h, w(p + "output.dense.weight"), w(p + "output.dense.bias")), self.p)
            x = layer_norm(h + x, w(p + "output.LayerNorm.weight"), w(p + "output.LayerNorm.bias"), 1e-12)
        return x

    def cls_head(self, x):
        """BERTPredictionHead.forward, model/modeling.py:245-254 (decoder weight tied to word embeddings :241)"""
        w = self.w
        x = gelu_erf(F.linear(x, w("cls.dense.weight"), w("cls.dense.bias")))
        x = layer_norm(x, w("cls.layernorm.weight"), w("cls.layernorm.bias"), 1e-12)
        return F.linear(x, w("multimodal_encoder.embeddings.word_embeddings.weight"), w("cls.decoder.bias"))

    # ------------------------------------------------------------------------- host-side text helpers
    def caption_inputs(self, txt, mask_prob=0.6):
        """inputs and labels of the caption passes, model/pretrain.py:424-433 (= :807-816 in forward_cap_single): 'unimlm' masks 60 % of the
        tokens and predicts them; 'lm' feeds the tokens as they are and predicts the NEXT token at every position (label 0 = padding and
        the last position -> ignored)"""
        if self.caption_type == "unimlm" and self.full_masker:                                   # full_mask, model/pretrain.py:137-142
            n = txt.shape[1]
            tokens = torch.cat((txt, torch.full_like(txt, self.mask_token)), dim=1)
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

    def text_masker(self, tokens, mask_prob):
        """TokenMasker.perform_mask, model/modeling.py:134-174 (CPU numpy + python `random`, >= 1 mask per row)."""
        tokens = np.array(tokens.cpu().numpy())
        ind = np.zeros(tokens.shape, dtype=np.int64)
        for i in range(len(ind)):
            while all(ind[i] == 0):
                for j in range(1, len(ind[0])):
                    if tokens[i][j] != 0 and random.random() < mask_prob:
                        ind[i][j] = 1
        labels = -np.ones(tokens.shape, dtype=np.int64)
        rng = list(range(*self.masker_range))
        for i in range(tokens.shape[0]):
            for j in range(tokens.shape[1]):
                if ind[i][j] == 1:
                    src = tokens[i][j]
                    prob = random.random()
                    if prob < 0.8:
                        tokens[i][j] = self.mask_token
                    elif prob < 0.9:
                        tokens[i][j] = random.choice(rng)
                    labels[i][j] = src
        return torch.from_numpy(tokens).long(), torch.from_numpy(labels).long()

    def get_task_prompt(self, sentence, batch_size):
        """VALORModel.get_task_prompt, model/modeling.py:355-369 (bert tokenizer branch): whole-word vocab lookup."""
        ids = [101] + [self.vocab.get(t, 100) for t in sentence.lower().split()] + [102]
        return torch.tensor(ids).unsqueeze(0).expand(batch_size, -1).long()

    # ------------------------------------------------------------------------- contrastive
    @staticmethod
    def compute_fine_matrix(featA, featB, maskA, maskB, weightA, weightB):
        """VALOR.compute_fine_matrix_slice, model/pretrain.py:191-211"""
        weightA = weightA.masked_fill((1 - maskA).bool(), float("-inf"))
        weightA = torch.softmax(weightA, dim=-1)
        weightB = weightB.masked_fill((1 - maskB).bool(), float("-inf"))
        weightB = torch.softmax(weightB, dim=-1)
        logits = torch.einsum("atd,bvd->abtv", featA, featB)
        logits = torch.einsum("abtv,at->abtv", logits, maskA.to(logits.dtype))
        logits = torch.einsum("abtv,bv->abtv", logits, maskB.to(logits.dtype))
        a2b = logits.max(dim=-1)[0]
        b2a = logits.max(dim=-2)[0]
        a2b = torch.einsum("abt,at->ab", a2b, weightA)
        b2a = torch.einsum("abv,bv->ab", b2a, weightB)
        return (a2b + b2a) / 2.0

    def contrastive_loss(self, score):
        """VALORModel.contrastive_loss, model/modeling.py:418-433 (clip video encoder: temp = 1/exp(logit_scale), else contra_temp)"""
        temp = 1.0 / self.w("clip_model.logit_scale").exp() if self.spec.video_encoder == "clip" else self.w("contra_temp")
        s = score / temp
        l1 = (-F.log_softmax(s, dim=1)).diag()
        l2 = (-F.log_softmax(s, dim=0)).diag()
        return torch.mean(torch.cat((l1, l2), dim=0))

    def fine_weight(self, name, feat):
        """nn.Sequential(Linear, ReLU, Linear(->1)), model/pretrain.py:104-116"""
        w = self.w
        h = F.relu(F.linear(feat, w(f"{name}_fine_weight.0.weight"), w(f"{name}_fine_weight.0.bias")))
        return F.linear(h, w(f"{name}_fine_weight.2.weight"), w(f"{name}_fine_weight.2.bias")).squeeze(2)

    # ------------------------------------------------------------------------- BASELINE configs[0]
    def text_mlm(self, bert_tokens, compute_loss=True):
        """The reference's CPU-runnable plumbing case (forward_pt cannot express text-only MLM): TokenMasker
        p = 0.15 (modeling.py:134-174) -> multimodal_encoder(txt, None, None, None, casual=False) (bert.py:848-896, no
        cross-attention input) -> BERTPredictionHead on the masked rows (modeling.py:245-254) -> F.cross_entropy."""
        txt_input, txt_labels = self.text_masker(bert_tokens, 0.15)
        o = self.bert_model(txt_input, None, None, None, False)
        scores = self.cls_head(o[txt_labels != -1])
        if compute_loss:
            return {"mlm_loss": F.cross_entropy(scores, txt_labels[txt_labels != -1])}
        return {"mlm_scores_t": scores, "txt_labels_mlm": txt_labels}

    def multimodal_inputs(self, video_output, audio_output, bs):
        """get_multimodal_forward_input_video / _audio, model/modeling.py:485-502"""
        w = self.w
        video_input = audio_input = None
        if video_output is not None:                                                             # modeling.py:485-493
            if "hidden_trans_video_multimodal.0.weight" in self.sd:                              # modeling.py:348-349,487-488
                video_output = layer_norm(F.linear(video_output, w("hidden_trans_video_multimodal.0.weight"), w("hidden_trans_video_multimodal.0.bias")),
                                          w("hidden_trans_video_multimodal.1.weight"), w("hidden_trans_video_multimodal.1.bias"), 1e-12)
            vo = video_output + w("video_frame_embedding")[:, :video_output.shape[1], :].unsqueeze(-2)
            video_input = vo.reshape(bs, -1, self.spec.hidden) + w("video_type_embeddings")
        if audio_output is not None:                                                             # modeling.py:495-502
            if "hidden_trans_audio_multimodal.0.weight" in self.sd:                              # modeling.py:350-351,497-498
                audio_output = layer_norm(F.linear(audio_output, w("hidden_trans_audio_multimodal.0.weight"), w("hidden_trans_audio_multimodal.0.bias")),
                                          w("hidden_trans_audio_multimodal.1.weight"), w("hidden_trans_audio_multimodal.1.bias"), 1e-12)
            ao = audio_output + w("audio_frame_embedding")[:, :audio_output.shape[1], :].unsqueeze(-2)
            audio_input = ao.reshape(bs, -1, self.spec.hidden) + w("audio_type_embeddings")
        return video_input, audio_input

    # ------------------------------------------------------------------------- the hot path
    def forward_pt(self, batch, task, compute_loss=True, gather=None, collect=None):
        """VALOR.forward_pt, model/pretrain.py:214-541 (contra_type='fine', caption_type='unimlm', va_concate).
        gather: optional callables (feat -> gathered feat, tokens -> gathered tokens) emulating
        ddp_allgather_with_grads / ddp_allgather (utils/distributed.py:38-93) for multi-rank tests."""
        w = self.w
        mlm_task, caption_task, contra_task = [], [], []
        for i in task.split("_"):
            if "mlm" in i:
                mlm_task = i.split("%")[1:]
            elif "caption" in i:
                caption_task = i.split("%")[1:]
            elif "contra" in i:
                contra_task = i.split("%")[1:]
        out = {}
        col = collect if collect is not None else {}
        txt_tokens = batch["txt_tokens"]
        alltasks = "".join(mlm_task + caption_task + contra_task)
        video_output = audio_output = None
        if "v" in alltasks:
            video_output = self.forward_video_encoder(batch["video_pixels"])
            col["video_output"] = video_output
        if "a" in alltasks:
            audio_output = self.forward_audio_encoder(batch["audio_spectrograms"])
            col["audio_output"] = audio_output
        if "t" in "".join(contra_task):
            if self.spec.txt_encoder == "bert":                        # pretrain.py:252-263, modeling.py:439-440 (casual=False)
                txt_tokens_contra = txt_tokens["bert_tokens"]
                prompt = self.get_task_prompt("project language in common space", txt_tokens_contra.shape[0]) if self.use_task_prompt else None
                txt_output = self.bert_model(txt_tokens_contra, prompt, None, None, False)[:, :txt_tokens_contra.shape[1]]
            else:
                txt_tokens_contra = txt_tokens["clip_tokens"]
                txt_output = self.clip_text(txt_tokens_contra)
            col["txt_output"] = txt_output

        coarse = self.spec.contra_type == "coarse"
        if contra_task and coarse:
            # contra_type 'coarse' (pool_*_for_contra, model/modeling.py:373-407): ONE vector per modality -- text: [CLS] (BERT) / the EOT row
            # (CLIP: tokens.argmax); video: the frames' [CLS] rows (Swin: token means) averaged over the frames; audio likewise -- BEFORE the heads
            if "t" in "".join(contra_task):
                txt_output = txt_output[:, 0] if self.spec.txt_encoder == "bert" else \
                    txt_output[torch.arange(txt_tokens_contra.shape[0]), txt_tokens_contra.argmax(dim=-1)]
            if "v" in "".join(contra_task):
                vo = video_output.mean(dim=2) if self.spec.video_encoder == "swin" else video_output[:, :, 0]
                video_output_c = vo.mean(dim=1)
            if "a" in "".join(contra_task):
                audio_output_c = audio_output[:, :, 0].mean(dim=1)
        if contra_task:
            feat_t = feat_v = feat_a = None
            if "t" in "".join(contra_task):
                if self.spec.txt_encoder == "bert":                                              # Contra_head, pretrain.py:33-38,94-95
                    feat_t = F.normalize(F.linear(txt_output, w("contra_head_t.linear.weight")), dim=-1)
                else:
                    feat_t = F.normalize(txt_output @ w("clip_model.text_projection"), dim=-1)   # pretrain.py:90,274-276
                if compute_loss and gather:
                    feat_t = gather[0](feat_t); txt_tokens_contra = gather[1](txt_tokens_contra)
            if "v" in "".join(contra_task):
                if coarse:
                    vp = video_output_c
                    feat_v = F.normalize(vp @ w("clip_model.visual.proj") if self.spec.clip_heads else F.linear(vp, w("contra_head_v.linear.weight")), dim=-1)
                elif self.spec.video_encoder == "swin":                                          # modeling.py:388-389 mean over tokens
                    feat_v = F.normalize(F.linear(video_output.mean(dim=2), w("contra_head_v.linear.weight")), dim=-1)
                elif self.spec.clip_heads:
                    feat_v = F.normalize(video_output[:, :, 0] @ w("clip_model.visual.proj"), dim=-1)  # :91, modeling.py:387
                else:            # CLIP video encoder beside a BERT text encoder (config/pretrain-VALOR-large.json): Contra_head, pretrain.py:93-97
                    feat_v = F.normalize(F.linear(video_output[:, :, 0], w("contra_head_v.linear.weight")), dim=-1)
                if compute_loss and gather:
                    feat_v = gather[0](feat_v)
            if "a" in "".join(contra_task):
                feat_a = F.normalize(F.linear(audio_output_c if coarse else audio_output[:, :, 0], w("contra_head_a.linear.weight")), dim=-1)
                if compute_loss and gather:
                    feat_a = gather[0](feat_a)
            col.update(feat_t=feat_t, feat_v=feat_v, feat_a=feat_a)
            if compute_loss and coarse:                                                          # pretrain.py:375-395
                losses = []
                if "tv" in contra_task:
                    losses.append(self.contrastive_loss(feat_t @ feat_v.t()))
                if "tva" in contra_task:
                    if self.spec.late_fusion:
                        sm = feat_t @ feat_v.t() + feat_t @ feat_a.t()
                    else:
                        feat_va = F.normalize(F.linear(torch.cat((feat_v, feat_a), dim=-1), w("va_fusion.weight"), w("va_fusion.bias")), dim=-1)
                        sm = feat_t @ feat_va.t()
                    col["score_tva"] = sm
                    losses.append(self.contrastive_loss(sm))
                if "ta" in contra_task:
                    losses.append(self.contrastive_loss(feat_t @ feat_a.t()))
                out["contra_loss"] = sum(losses) / len(losses) * self.contra_loss_ratio
            elif compute_loss:
                losses = []
                maskA = (txt_tokens_contra != 0).long() if feat_t is not None else None
                if "tva" in contra_task and self.spec.late_fusion:                               # pretrain.py:313-321: unit token weights, two matrices summed
                    ones_w = lambda f: torch.ones_like(f[:, :, 0])
                    sm = self.compute_fine_matrix(feat_t, feat_v, maskA, torch.ones(*feat_v.shape[:2]).long(), ones_w(feat_t), ones_w(feat_v)) + \
                        self.compute_fine_matrix(feat_t, feat_a, maskA, torch.ones(*feat_a.shape[:2]).long(), ones_w(feat_t), ones_w(feat_a))
                    col["score_tva"] = sm
                    losses.append(self.contrastive_loss(sm))
                elif "tva" in contra_task:                                                       # pretrain.py:311-336
                    feat_va = torch.cat((feat_v, feat_a), dim=1)
                    maskB = torch.ones(*feat_va.shape[:2]).long()
                    wA = self.fine_weight("text", feat_t)
                    wB = torch.cat((self.fine_weight("video", feat_v), self.fine_weight("audio", feat_a)), dim=1)
                    sm = self.compute_fine_matrix(feat_t, feat_va, maskA, maskB, wA, wB)
                    col["score_tva"] = sm
                    losses.append(self.contrastive_loss(sm))
                if "tv" in contra_task:                                                          # :303-309
                    maskB = torch.ones(*feat_v.shape[:2]).long()
                    sm = self.compute_fine_matrix(feat_t, feat_v, maskA, maskB, self.fine_weight("text", feat_t),
                                                  self.fine_weight("video", feat_v))
                    col["score_tv"] = sm
                    losses.append(self.contrastive_loss(sm))
                if "ta" in contra_task:                                                          # :339-345
                    maskB = torch.ones(*feat_a.shape[:2]).long()
                    sm = self.compute_fine_matrix(feat_t, feat_a, maskA, maskB, self.fine_weight("text", feat_t),
                                                  self.fine_weight("audio", feat_a))
                    col["score_ta"] = sm
                    losses.append(self.contrastive_loss(sm))
                ones = lambda f: torch.ones(*f.shape[:2]).long()
                if "va" in contra_task:                                                          # :346-352
                    sm = self.compute_fine_matrix(feat_v, feat_a, ones(feat_v), ones(feat_a), self.fine_weight("video", feat_v),
                                                  self.fine_weight("audio", feat_a))
                    col["score_va"] = sm
                    losses.append(self.contrastive_loss(sm))
                if "vta" in contra_task:                                                         # :354-361
                    sm = self.compute_fine_matrix(feat_v, torch.cat((feat_t, feat_a), dim=1), ones(feat_v), torch.cat((maskA, ones(feat_a)), dim=1),
                                                  self.fine_weight("video", feat_v),
                                                  torch.cat((self.fine_weight("text", feat_t), self.fine_weight("audio", feat_a)), dim=1))
                    col["score_vta"] = sm
                    losses.append(self.contrastive_loss(sm))
                if "atv" in contra_task:                                                         # :363-370
                    sm = self.compute_fine_matrix(feat_a, torch.cat((feat_t, feat_v), dim=1), ones(feat_a), torch.cat((maskA, ones(feat_v)), dim=1),
                                                  self.fine_weight("audio", feat_a),
                                                  torch.cat((self.fine_weight("text", feat_t), self.fine_weight("video", feat_v)), dim=1))
                    col["score_atv"] = sm
                    losses.append(self.contrastive_loss(sm))
                out["contra_loss"] = sum(losses) / len(losses) * self.contra_loss_ratio
            else:
                out.update(feat_t=feat_t, feat_v=feat_v, feat_a=feat_a, txt_tokens=txt_tokens_contra)

        txt = txt_tokens["bert_tokens"]
        bs = txt.shape[0]
        video_input, audio_input = self.multimodal_inputs(video_output, audio_output, bs)

        def run_group(txt_input, txt_labels, prompt, g, casual, tag):
            vi = video_input if "v" in g else None
            ai = audio_input if "a" in g else None
            o = self.bert_model(txt_input, prompt, vi, ai, casual)[:, :txt_input.shape[1], :]
            o = o[txt_labels != -1]
            scores = self.cls_head(o)
            col[f"{tag}_scores_{g}"] = scores
            if compute_loss:
                return F.cross_entropy(scores, txt_labels[txt_labels != -1])
            out[f"{tag}_scores_{g}"] = scores
            return None

        if caption_task:                                                                         # pretrain.py:419-481
            if self.full_masker:      # the reference slices the 'tv' / 'ta' outputs with the original length against the doubled labels (:454): IndexError
                raise NotImplementedError("full_masker with a pretraining caption task fails in the reference (model/pretrain.py:454)")
            txt_input, txt_labels = self.caption_inputs(txt)
            col["caption_txt_input"], col["caption_txt_labels"] = txt_input, txt_labels
            lo = []
            for g in ("tva", "tv", "ta"):
                if g in caption_task:
                    prompt = self.get_task_prompt("describe the video with natural language", bs) if self.use_task_prompt else None
                    l = run_group(txt_input, txt_labels, prompt, g, True, "caption")
                    if l is not None:
                        lo.append(l)
            if compute_loss:
                out["caption_loss"] = sum(lo) / len(lo)
            else:
                out["txt_labels_caption"] = txt_labels
        if mlm_task:                                                                             # pretrain.py:483-535
            txt_input, txt_labels = self.text_masker(txt, 0.15)
            col["mlm_txt_input"], col["mlm_txt_labels"] = txt_input, txt_labels
            sent = {"tva": "predict masked tokens with visual and audio cues", "tv": "predict masked tokens with visual cues",
                    "ta": "predict masked tokens with audio cues"}
            lo = []
            for g in ("tva", "tv", "ta"):
                if g in mlm_task:
                    l = run_group(txt_input, txt_labels, self.get_task_prompt(sent[g], bs), g, False, "mlm")
                    if l is not None:
                        lo.append(l)
            if compute_loss:
                out["mlm_loss"] = sum(lo) / len(lo)
            else:
                out["txt_labels_mlm"] = txt_labels
        return out


    # ------------------------------------------------------------------------- finetune / inference tasks
    def forward(self, batch, task, compute_loss=True, **kw):
        """VALOR.forward, model/pretrain.py:125-135"""
        if task.startswith("pt"):
            return self.forward_pt(batch, task, compute_loss, **kw)
        if task.startswith("ret"):
            return self.forward_ret(batch, task, compute_loss, **kw)
        if task.startswith("cap"):
            return self.forward_cap(batch, task, compute_loss)
        if task.startswith("qa"):
            return self.forward_qa(batch, task, compute_loss)
        raise NotImplementedError(task)

    def forward_ret(self, batch, task, compute_loss=True, gather=None):
        """VALOR.forward_ret, model/pretrain.py:544-711. Line by line the contrastive branch of forward_pt (:252-407: same encoders, pooling,
        heads, gathers, fine matrices and InfoNCE) on the groups after 'ret%', except that the mean of the group losses is NOT scaled by
        contra_loss_ratio (:706 vs :406); compute_loss=False returns feat_t / feat_v / feat_a / txt_tokens (:708-715)."""
        ratio, self.contra_loss_ratio = self.contra_loss_ratio, 1.0
        try:
            return self.forward_pt(batch, "pt_contra%" + "%".join(task.split("%")[1:]), compute_loss, gather)
        finally:
            self.contra_loss_ratio = ratio

    def forward_cap(self, batch, task, compute_loss=True, beam_size=3, max_generation_len=30):
        """VALOR.forward_cap -> forward_cap_single (loss) / generate_cap, model/pretrain.py:713-725,794-985 (caption_type 'unimlm', no
        label smoothing, no scst, full_masker False: the shipped caption-*.json settings)."""
        groups = task.split("%")[1:]
        txt = batch["txt_tokens"]["bert_tokens"] if batch.get("txt_tokens") is not None else None
        alltasks = "".join(groups)
        video_output = self.forward_video_encoder(batch["video_pixels"]) if "v" in alltasks else None
        audio_output = self.forward_audio_encoder(batch["audio_spectrograms"]) if "a" in alltasks else None
        bs = (video_output if video_output is not None else audio_output).shape[0]
        video_input, audio_input = self.multimodal_inputs(video_output, audio_output, bs)
        if compute_loss:                                                                         # :802-880
            txt_input, txt_labels = self.caption_inputs(txt)
            lo = []
            for g in ("tva", "tv", "ta"):
                if g in groups:
                    prompt = self.get_task_prompt("describe the video with natural language", bs) if self.use_task_prompt else None
                    o = self.bert_model(txt_input, prompt, video_input if "v" in g else None, audio_input if "a" in g else None, True,
                                        self.full_masker)                                        # :835,847,859
                    scores = self.cls_head(o[:, :txt_input.shape[1]][txt_labels != -1])
                    if self.label_smoothing > 0:          # LabelSmoothing, model/pretrain.py:46-61: KL(smoothed target || softmax), summed over the vocabulary, mean over rows
                        logp = F.log_softmax(scores, dim=-1)
                        tgt = torch.full_like(logp, self.label_smoothing / (logp.shape[1] - 1))
                        tgt.scatter_(1, txt_labels[txt_labels != -1].unsqueeze(1), 1.0 - self.label_smoothing)
                        lo.append((tgt * (tgt.log() - logp)).sum(1).mean())
                    else:
                        lo.append(F.cross_entropy(scores, txt_labels[txt_labels != -1]))
            return {"caption_loss": sum(lo) / len(lo)}
        ev = {}                                                                                  # generate_cap :914-985
        for g, key in (("tv", "t_v"), ("tva", "t_va"), ("ta", "t_a")):
            if g in groups:
                prompt = self.get_task_prompt("describe the video with natural language", bs) if self.use_task_prompt else None
                vi, ai = (video_input if "v" in g else None), (audio_input if "a" in g else None)
                if beam_size > 1:
                    ev["generated_sequences_" + key] = self.decode_beam(vi, ai, prompt, bs, beam_size, max_generation_len)
                else:
                    ev["generated_sequences_" + key], ev["logprobs_" + key] = self.decode_greedy(vi, ai, prompt, bs, max_generation_len)
        return ev

    def qa_prompt(self, question_tokens):
        """the 'task prompt' of the QA passes is the QUESTION (prompt-type embeddings), with 'answer the question' spliced in behind its
        [CLS] when use_task_prompt, model/pretrain.py:1268-1274"""
        if not self.use_task_prompt:
            return question_tokens
        tp = self.get_task_prompt("answer the question", question_tokens.shape[0])[:, 1:-1]
        return torch.cat((question_tokens[:, 0:1], tp, question_tokens[:, 1:]), dim=1)

    def forward_qa(self, batch, task, compute_loss=True, beam_size_qa=1, max_generation_len=30):
        """VALOR.forward_qa -> forward_qa_single (loss) / generate_qa, model/pretrain.py:1191-1459. Loss: one answer per question (the
        video-QA datasets) or several weighted candidates (image QA: question / video / audio rows tiled per answer, :1243-1265, loss
        rows weighted and summed over the QUESTION count :1288-1290); generation: sample_num[i] questions of clip i (:1378-1390).
        Loss :1276-1290: TokenMasker p = 0.99 on the answer tokens, per-SAMPLE mean of the masked-token CE, mean over samples, mean over groups."""
        groups = task.split("%")[1:]
        q = batch["question_tokens"]["bert_tokens"]
        alltasks = "".join(groups)
        video_output = self.forward_video_encoder(batch["video_pixels"]) if "v" in alltasks else None
        audio_output = self.forward_audio_encoder(batch["audio_spectrograms"]) if "a" in alltasks else None
        bs = q.shape[0]
        clips = (video_output if video_output is not None else audio_output).shape[0]
        video_input, audio_input = self.multimodal_inputs(video_output, audio_output, clips)
        sn = [int(n) for n in batch.get("sample_num", [1] * clips)]
        if not compute_loss and any(n != 1 for n in sn):      # generate_qa :1378-1390: clip i serves sample_num[i] consecutive question rows
            idx = torch.tensor([i for i, n in enumerate(sn) for _ in range(n)])
            video_input = video_input[idx] if video_input is not None else None
            audio_input = audio_input[idx] if audio_input is not None else None
        prompt = self.qa_prompt(q)
        if compute_loss:
            nums = [int(n) for n in batch["answer_nums"]]
            tile = not all(n == 1 for n in nums)                      # image QA: several weighted candidate answers per question, :1243-1265
            if tile:
                rep = torch.tensor(nums)
                prompt = self.qa_prompt(q.repeat_interleave(rep, dim=0))
                video_input = video_input.repeat_interleave(rep, dim=0) if video_input is not None else None
                audio_input = audio_input.repeat_interleave(rep, dim=0) if audio_input is not None else None
            txt_input, txt_labels = self.caption_inputs(batch["txt_tokens"]["bert_tokens"], 0.99)
            lo = []
            for g in ("tva", "tv", "ta"):
                if g in groups:
                    o = self.bert_model(txt_input, prompt, video_input if "v" in g else None, audio_input if "a" in g else None, True,
                                        self.full_masker)                                        # :1276,1300,1324
                    scores = self.cls_head(o[:, :txt_input.shape[1]])
                    b, n, c = scores.shape
                    loss = F.cross_entropy(scores.reshape(b * n, c), txt_labels.reshape(b * n), ignore_index=-1, reduction="none").reshape(b, n)
                    loss = loss.sum(dim=-1) / (txt_labels != -1).sum(dim=-1)
                    lo.append((loss * torch.as_tensor(batch["answer_weights"], dtype=loss.dtype)).sum() / len(nums) if tile else loss.mean())
            return {"qa_loss": sum(lo) / len(lo)}
        ev = {}                                                                                  # generate_qa :1366-1459
        for g, key in (("tv", "t_v"), ("tva", "t_va"), ("ta", "t_a")):
            if g in groups:
                vi, ai = (video_input if "v" in g else None), (audio_input if "a" in g else None)
                if beam_size_qa > 1:
                    ev["generated_answers_" + key] = self.decode_beam(vi, ai, prompt, bs, beam_size_qa, max_generation_len)
                else:
                    ev["generated_answers_" + key] = self.decode_greedy(vi, ai, prompt, bs, max_generation_len)[0]
        return ev

    BOS, EOS, MASK = 101, 102, 103        # [CLS] / [SEP] / [MASK] of bert-base-uncased, model/modeling.py:669-671

    def get_logits(self, state, vi, ai, prompt, rows, trace=None):
        """VALOR.get_logits (unimlm) + forward_cap_single(compute_loss=False), model/pretrain.py:1031-1051,882-900: the decoder runs on
        [CLS] + generated tokens + [MASK] from scratch every step (multimodal_use_cross_attn: no cache), logits of the last text position."""
        mask_col = torch.full((rows, 1), self.MASK, dtype=torch.long)
        bos = torch.full((rows, 1), self.BOS, dtype=torch.long)
        if self.caption_type == "lm":                                                            # :1038-1040: no [MASK], the last token's logits
            txt = torch.cat((bos, state), dim=1) if state is not None else bos
        else:
            txt = torch.cat((bos, state, mask_col), dim=1) if state is not None else torch.cat((bos, mask_col), dim=1)
        o = self.bert_model(txt, prompt, vi, ai, True)[:, :txt.shape[1]]
        logits = self.cls_head(o[:, -1])
        if trace is not None:
            trace.append(logits)
        return logits

    def decode_greedy(self, vi, ai, prompt, bs, max_len, trace=None):
        """VALOR.decode_greedy, model/pretrain.py:988-1028 (mode 'greedy'; logprobs stay zero in that mode)"""
        sents = torch.full((bs, max_len), self.EOS, dtype=torch.long)
        logprobs = torch.zeros(bs, max_len)
        unfinished = torch.ones(bs, dtype=torch.bool)
        state = None
        for t in range(max_len):
            logits = self.get_logits(state, vi, ai, prompt, bs, trace)
            wt = logits.max(1)[1].view(-1).long()
            unfinished = unfinished * (wt != self.EOS)
            wt = wt * unfinished.type_as(wt) + (1 - unfinished.type_as(wt)) * self.EOS
            sents[:, t] = wt
            state = wt.unsqueeze(1) if state is None else torch.cat((state, wt.unsqueeze(1)), dim=1)
            if unfinished.sum() == 0:
                break
        return sents, logprobs

    def decode_beam(self, vi, ai, prompt, bs, beam, max_len, trace=None, gaps=None):
        """VALOR.decode_beam / select / _adjust_tensor / expand_tensor, model/pretrain.py:1054-1189. Rows are ordered (sample, beam)."""
        seq_logprob = torch.zeros(bs, 1, 1)
        seq_mask = torch.ones(bs, beam, 1)
        outputs, selected_words, state = [], None, None
        expand = lambda x: None if x is None else x.unsqueeze(1).expand(-1, beam, *x.shape[1:]).reshape(-1, *x.shape[1:])
        for t in range(max_len):
            cur = 1 if t == 0 else beam
            logits = self.get_logits(state, vi, ai, prompt, bs * cur, trace)
            word_logprob = F.log_softmax(logits, dim=1).view(bs, cur, -1)
            cand = seq_logprob + word_logprob
            if t > 0:                                                                            # a beam that met EOS keeps its score
                mask = (selected_words.view(bs, cur) != self.EOS).float().unsqueeze(-1)
                seq_mask = seq_mask * mask
                old = seq_logprob.expand_as(cand).contiguous()
                cand = seq_mask * cand + old * (1 - seq_mask)
            sel_logprob, sel_idx = torch.sort(cand.view(bs, -1), -1, descending=True)            # select :1156-1159
            if gaps is not None:                 # smallest gap among the best beam + 1 candidates: what a lower-precision run must resolve
                gaps.append((sel_logprob[:, :beam] - sel_logprob[:, 1:beam + 1]).min(dim=1).values)
            sel_logprob, sel_idx = sel_logprob[:, :beam], sel_idx[:, :beam]
            V = cand.shape[-1]
            sel_beam = sel_idx // V
            selected_words = sel_idx - sel_beam * V
            seq_logprob = sel_logprob.unsqueeze(-1)
            seq_mask = torch.gather(seq_mask, 1, sel_beam.unsqueeze(-1))
            outputs = [torch.gather(o, 1, sel_beam.unsqueeze(-1)) for o in outputs]
            outputs.append(selected_words.unsqueeze(-1))
            selected_words = selected_words.view(-1, 1)
            if state is not None:                                                                # _adjust_tensor, 2-d case
                state = torch.gather(state.view(bs, beam, -1), 1, sel_beam.unsqueeze(-1).expand(bs, beam, state.shape[1])).reshape(bs * beam, -1)
                state = torch.cat((state, selected_words), dim=1)
            else:
                state = selected_words
            if t == 0:                                                                           # expand_tensor :1133-1139
                vi, ai, prompt = expand(vi), expand(ai), expand(prompt)
        seq_logprob, sort_idx = torch.sort(seq_logprob, 1, descending=True)
        outputs = torch.cat(outputs, -1)
        outputs = torch.gather(outputs, 1, sort_idx.expand(bs, beam, max_len))
        return outputs.contiguous()[:, 0]

# ----------------------------------------------------------------------------- state-dict helpers
def is_alias_key(k):
    """keys of the reference state dict that share storage with another key: the tied decoder weight (modeling.py:241) and the
    txt_encoder.* view of the shared multimodal encoder (modeling.py:689-691)"""
    return k == "cls.decoder.weight" or k.startswith("txt_encoder.")


def trainable_copy(sd):
    """leaf copies (requires_grad) of a synthetic state dict for the oracle; aliases point at their owners, integer buffers
    (relative_position_index) are passed through"""
    out = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items() if not is_alias_key(k)}
    out["cls.decoder.weight"] = out["multimodal_encoder.embeddings.word_embeddings.weight"]
    for k in sd:
        if k.startswith("txt_encoder."):
            out[k] = out["multimodal_encoder." + k[len("txt_encoder."):]]
    return out


# ----------------------------------------------------------------------------- optimizer
def param_group_of(name, new_params_name=()):
    """optim/misc.py:13-64 group assignment -> (group_index 0..9, decayed?)."""
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    nd = any(t in name for t in no_decay)
    if "clip" in name and "visual" in name:
        base = 4
    elif "clip" in name:
        base = 6
    elif "multimodal_encoder.decoder" in name:
        base = 8
    elif any(t in name for t in new_params_name):
        base = 2
    else:
        base = 0
    return base + (1 if nd else 0)


def group_hparams(learning_rate, weight_decay, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=0.0, decoder_lr=-1):
    """optim/misc.py:66-77 ; train_utils.py:614-615 defaults."""
    if decoder_lr == -1:
        decoder_lr = learning_rate
    lrs = [learning_rate, learning_rate, new_lr, new_lr, clip_lr, clip_lr, clip_lr_text, clip_lr_text, decoder_lr, decoder_lr]
    wds = [weight_decay if i % 2 == 0 else 0.0 for i in range(10)]
    return lrs, wds


def warmup_linear(x, warmup_ratio):
    """optim/sched.py:27-34"""
    if x < warmup_ratio:
        return x / warmup_ratio
    return max((x - 1.0) / (warmup_ratio - 1.0), 0)


def adamw_step(params, grads, state, lrs, wds, groups, betas=(0.9, 0.98), eps=1e-6):
    """optim/adamw.py:40-103 on dicts name -> tensor (in place). state[name] = dict(step, exp_avg, exp_avg_sq)."""
    b1, b2 = betas
    for name, p in params.items():
        g = grads.get(name)
        if g is None:
            continue
        st = state.setdefault(name, {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)})
        st["step"] += 1
        st["exp_avg"].mul_(b1).add_(g, alpha=1.0 - b1)
        st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = st["exp_avg_sq"].sqrt().add_(eps)
        lr, wd = lrs[groups[name]], wds[groups[name]]
        step_size = lr * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
        p.addcdiv_(st["exp_avg"], denom, value=-step_size)
        if wd > 0.0:
            p.add_(p, alpha=-lr * wd)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (train_utils.py:358-360). Returns total norm; scales in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = min(1.0, float(max_norm / (total + 1e-6)))
    for g in grads.values():
        g.mul_(coef)
    return total
