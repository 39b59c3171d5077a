"""CPU emulation of bf16 storage policies for the contrastive branch (TEST / DESIGN TOOL, never imported by the product).

Question it answers: which tensors must stay fp32 for the bf16 model's contra_loss to land within
1e-3 of the fp32 reference? The emulator re-runs the oracle's encoders with a rounding function applied where the native
path stores bf16 (GEMM operands / outputs, LayerNorm outputs, the residual stream, features), under different policies:

  native   : everything the HIP path stores is bf16 (residual stream included)            -- round 1 behaviour
  res32    : the pre-LN residual stream x stays fp32 (GEMM operands / LN outputs still bf16)
  head32   : native + fp32 from the last LayerNorm on (projection, L2 normalise, similarity)
  res32+head32

usage: python tools/precision_emulator.py [tiny|base] [B] [F] [A]
"""
import math
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import valor_oracle as VO          # noqa: E402
from valor_amd import synth        # noqa: E402


def r16(t):
    return t.bfloat16().float()


class Emu(VO.Oracle):
    def __init__(self, *a, res32=False, head32=False, **k):
        super().__init__(*a, **k)
        self.res32, self.head32 = res32, head32
        self.rr = (lambda t: t) if res32 else r16       # residual-stream storage

    def lin(self, x, wk, bk=None, act=None):
        w = r16(self.w(wk))
        y = F.linear(r16(x), w, None)
        if bk is not None:
            y = y + r16(self.w(bk))
        if act is not None:
            y = act(y)
        return r16(y)

    def ln(self, x, wk, bk, eps):
        return r16(VO.layer_norm(x, r16(self.w(wk)), r16(self.w(bk)), eps))

    def attn(self, qkv, heads, mask):
        E = qkv.shape[-1] // 3
        q, k, v = qkv.split(E, dim=-1)
        N, L, _ = q.shape
        hd = E // heads
        q = q.view(N, L, heads, hd).transpose(1, 2)
        k = k.view(N, L, heads, hd).transpose(1, 2)
        v = v.view(N, L, heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
        if mask is not None:
            s = s + mask[:, None]
        a = r16(torch.softmax(s, dim=-1)) @ v
        return r16(a.transpose(1, 2).reshape(N, L, E))

    def clip_block(self, x, p, heads, mask):
        h = self.ln(x, p + "ln_1.weight", p + "ln_1.bias", 1e-5)
        a = self.attn(self.lin(h, p + "attn.in_proj_weight", p + "attn.in_proj_bias"), heads, mask)
        o = self.lin(a, p + "attn.out_proj.weight")
        x = self.rr(x + o + r16(self.w(p + "attn.out_proj.bias")))
        h = self.ln(x, p + "ln_2.weight", p + "ln_2.bias", 1e-5)
        h = self.lin(h, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", VO.quick_gelu)
        o = self.lin(h, p + "mlp.c_proj.weight")
        return self.rr(x + o + r16(self.w(p + "mlp.c_proj.bias")))

    def final_ln(self, x, wk, bk, eps):
        y = VO.layer_norm(x, r16(self.w(wk)), r16(self.w(bk)), eps)
        return y if self.head32 else r16(y)

    def clip_visual(self, images):
        sp = self.spec
        x = F.conv2d(r16(images), r16(self.w("clip_model.visual.conv1.weight")), None, stride=sp.patch)
        x = r16(x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1))
        cls = r16(self.w("clip_model.visual.class_embedding")) + torch.zeros(x.shape[0], 1, x.shape[-1])
        x = r16(torch.cat([cls, x], dim=1) + r16(self.w("clip_model.visual.positional_embedding")))
        x = self.rr(VO.layer_norm(x, r16(self.w("clip_model.visual.ln_pre.weight")), r16(self.w("clip_model.visual.ln_pre.bias")), 1e-5))
        for i in range(sp.vis_layers):
            x = self.clip_block(x, f"clip_model.visual.transformer.resblocks.{i}.", sp.vis_heads, None)
        return self.final_ln(x, "clip_model.visual.ln_post.weight", "clip_model.visual.ln_post.bias", 1e-5)

    def clip_text(self, tokens):
        sp = self.spec
        x = r16(self.w("clip_model.token_embedding.weight"))[tokens]
        L = x.shape[1]
        x = self.rr(x + r16(self.w("clip_model.positional_embedding"))[:L])
        m = torch.tril((tokens != 0).long().unsqueeze(1).expand(-1, L, -1).clone())
        am = (1.0 - m.float()) * -10000.0
        for i in range(sp.txt_layers):
            x = self.clip_block(x, f"clip_model.transformer.resblocks.{i}.", sp.txt_heads, am)
        return self.final_ln(x, "clip_model.ln_final.weight", "clip_model.ln_final.bias", 1e-5)

    def forward_audio_encoder(self, audio):
        sp = self.spec
        b, n, hh, ww = audio.shape
        x = r16(audio.reshape(-1, hh, ww)).unsqueeze(1)
        x = F.conv2d(x, r16(self.w("audio_embeddings.first_conv.weight")), None, stride=sp.aud_patch)
        bb, c = x.shape[0], x.shape[1]
        x = r16(x.permute(0, 2, 3, 1).reshape(bb, -1, c)) + r16(self.w("audio_embeddings.first_conv.bias"))
        x = torch.cat((r16(self.w("audio_embeddings.cls_token")).expand(bb, -1, -1), x), dim=1)
        x = self.rr(x + r16(self.w("audio_embeddings.position_embeddings.weight"))[None])
        H = sp.aud_heads
        for i in range(sp.aud_layers):
            p = f"audio_encoder.layer.{i}."
            h = self.ln(x, p + "layernorm1.weight", p + "layernorm1.bias", 1e-12)
            qkv = torch.cat([self.lin(h, p + f"attention.linears.{j}.weight", p + f"attention.linears.{j}.bias") for j in range(3)], dim=-1)
            a = self.attn(qkv, H, None)
            o = self.lin(a, p + "attention.linears.3.weight")
            x = self.rr(x + o + r16(self.w(p + "attention.linears.3.bias")))
            h = self.ln(x, p + "layernorm2.weight", p + "layernorm2.bias", 1e-12)
            h = self.lin(h, p + "ff_layer.linear1.weight", p + "ff_layer.linear1.bias", VO.gelu_erf)
            o = self.lin(h, p + "ff_layer.linear2.weight")
            x = self.rr(x + o + r16(self.w(p + "ff_layer.linear2.bias")))
        x = self.final_ln(x, "audio_encoder.last_layernorm.weight", "audio_encoder.last_layernorm.bias", 1e-12)
        return x.reshape(b, n, -1, x.shape[-1])

    def contra_only(self, batch):
        """the contrastive branch of forward_pt with the head's storage policy"""
        w = self.w
        q = (lambda t: t) if self.head32 else r16
        vo = self.forward_video_encoder(batch["video_pixels"])
        ao = self.forward_audio_encoder(batch["audio_spectrograms"])
        tok = batch["txt_tokens"]["clip_tokens"]
        to = self.clip_text(tok)
        wq = (lambda k: w(k)) if self.head32 else (lambda k: r16(w(k)))
        feat_t = q(F.normalize(q(to @ wq("clip_model.text_projection")), dim=-1))
        feat_v = q(F.normalize(q(vo[:, :, 0] @ wq("clip_model.visual.proj")), dim=-1))
        feat_a = q(F.normalize(q(F.linear(ao[:, :, 0], wq("contra_head_a.linear.weight"))), dim=-1))
        maskA = (tok != 0).long()
        fw = lambda name, f: self.fine_weight(name, f)
        losses = []
        for g in ("tva", "tv", "ta"):
            fB = {"tva": torch.cat((feat_v, feat_a), dim=1), "tv": feat_v, "ta": feat_a}[g]
            wB = {"tva": torch.cat((fw("video", feat_v), fw("audio", feat_a)), dim=1), "tv": fw("video", feat_v), "ta": fw("audio", feat_a)}[g]
            sm = self.compute_fine_matrix(feat_t, fB, maskA, torch.ones(*fB.shape[:2]).long(), fw("text", feat_t), wB)
            losses.append(self.contrastive_loss(sm))
        return sum(losses) / 3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    Fr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    A = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    spec = synth.tiny_spec() if which == "tiny" else synth.base_spec()
    torch.manual_seed(0)
    for seed in (3, 7, 11):
        sd = synth.make_state_dict(spec, seed=seed, w_std=0.05 if which == "tiny" else 0.02)
        if os.environ.get("ROUND_W", "0") == "1":      # bf16-representable weights on both sides
            sd = {k: (r16(v) if v.is_floating_point() else v) for k, v in sd.items()}
        batch = synth.make_batch(spec, batch=B, frames=Fr, audio_slices=A, txt_len=32, seed=seed + 1)
        with torch.no_grad():
            orc = VO.Oracle(spec, sd, vocab_tokens=synth.synthetic_vocab(spec.vocab))
            random.seed(1)
            ref = float(orc.forward_pt(batch, "pt_contra%tva%tv%ta", compute_loss=True)["contra_loss"])
            line = [f"seed {seed}: ref {ref:.6f}"]
            for name, kw in (("native", {}), ("res32", dict(res32=True)), ("head32", dict(head32=True)),
                             ("res32+head32", dict(res32=True, head32=True))):
                e = Emu(spec, sd, vocab_tokens=synth.synthetic_vocab(spec.vocab), **kw)
                v = float(e.contra_only(batch))
                line.append(f"{name} {abs(v - ref) / abs(ref):.2e}")
            print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
