"""Where does the bf16 model's contrastive-loss error come from? (DIAGNOSTIC, GPU box; uses the oracle as the fp32 yardstick.)
For one `_q` golden fixture (bf16-representable weights / inputs):
  * encoder outputs / features of the native bf16 model vs the fp32 oracle (relative L2 error per tensor);
  * the contrastive loss re-computed in fp64 torch from (a) the reference features, (b) the native features, (c) mixtures with one
    tower swapped -- against the native kernel's loss: separates "the features are off" from "the head kernels are off".
usage: python tools/bf16_error_attribution.py [fixture name] [out.json]"""
import json
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import valor_oracle as VO                # noqa: E402
from valor_amd import synth              # noqa: E402
from valor_amd.model.valor import VALOR  # noqa: E402

TASK = "pt_contra%tva%tv%ta"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def head_fp64(sd, ft, fv, fa, tok):
    """pretrain.py:191-211 + modeling.py:418-433 in fp64 on given features"""
    d = lambda k: sd[k].double()
    fw = lambda n, f: (F.relu(F.linear(f, d(f"{n}_fine_weight.0.weight"), d(f"{n}_fine_weight.0.bias"))) @ d(f"{n}_fine_weight.2.weight").t()
                       + d(f"{n}_fine_weight.2.bias")).squeeze(-1)
    maskA = (tok != 0).long()
    k = d("clip_model.logit_scale").exp()
    losses = []
    for g in ("tva", "tv", "ta"):
        fB = {"tva": torch.cat((fv, fa), 1), "tv": fv, "ta": fa}[g]
        wB = {"tva": torch.cat((fw("video", fv), fw("audio", fa)), 1), "tv": fw("video", fv), "ta": fw("audio", fa)}[g]
        sm = VO.Oracle.compute_fine_matrix(ft, fB, maskA, torch.ones(*fB.shape[:2]).long(), fw("text", ft), wB)
        s = sm * k
        losses.append(torch.mean(torch.cat(((-F.log_softmax(s, 1)).diag(), (-F.log_softmax(s, 0)).diag()))))
    return float(sum(losses) / 3)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ref_base_b2f2a1_q"
    g = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
    rc = g["recipe"]
    spec = synth.ValorSpec(**rc["spec"])
    sd = synth.make_state_dict(spec, seed=rc["weight_seed"], bf16_exact=True)
    batch = synth.make_batch(spec, batch=rc["batch"], frames=rc["frames"], audio_slices=rc["audio_slices"], txt_len=rc["txt_len"],
                             seed=rc["batch_seed"], bf16_exact=True)
    out = {"fixture": name}
    col_o = {}
    with torch.no_grad():
        orc = VO.Oracle(spec, sd, vocab_tokens=synth.synthetic_vocab(spec.vocab))
        ref_loss = float(orc.forward_pt(batch, TASK, compute_loss=True, collect=col_o)["contra_loss"])
    for dt in (torch.bfloat16, torch.float32):
        model = VALOR({"dropout": 0.0}, spec=spec, dtype=dt, device="cuda:0")
        model.load_state_dict(sd, strict=True)
        model.train()
        model.collect = {}
        with torch.no_grad():
            nat_loss = float(model(batch, task=TASK, compute_loss=True)["contra_loss"])
        c = model.collect
        r = {"native_loss": nat_loss, "ref_loss": ref_loss, "rel_err": abs(nat_loss - ref_loss) / abs(ref_loss)}
        for k in ("video_output", "audio_output", "txt_output", "feat_t", "feat_v", "feat_a"):
            r["err_" + k] = rel(c[k], col_o[k].reshape(c[k].shape))
        tok = batch["txt_tokens"]["clip_tokens"]
        fr = {k: col_o[k].double() for k in ("feat_t", "feat_v", "feat_a")}
        fn = {k: c[k].double().cpu().reshape(fr[k].shape) for k in fr}
        h = lambda t, v, a: head_fp64(sd, t, v, a, tok)
        r["head64_on_ref_feats"] = h(fr["feat_t"], fr["feat_v"], fr["feat_a"])
        r["head64_on_native_feats"] = h(fn["feat_t"], fn["feat_v"], fn["feat_a"])
        r["head64_native_t_only"] = h(fn["feat_t"], fr["feat_v"], fr["feat_a"])
        r["head64_native_v_only"] = h(fr["feat_t"], fn["feat_v"], fr["feat_a"])
        r["head64_native_a_only"] = h(fr["feat_t"], fr["feat_v"], fn["feat_a"])
        # the reference features rounded to bf16 (what a perfect encoder would hand the bf16 head)
        rb = {k: v.float().bfloat16().double() for k, v in fr.items()}
        r["head64_on_rounded_ref_feats"] = h(rb["feat_t"], rb["feat_v"], rb["feat_a"])
        for k in list(r):
            if k.startswith("head64"):
                r[k + "_relerr"] = abs(r[k] - ref_loss) / abs(ref_loss)
        r["head_kernel_vs_head64_on_native_feats"] = abs(nat_loss - r["head64_on_native_feats"]) / abs(ref_loss)
        out[str(dt)] = r
        print(str(dt), json.dumps({k: (round(v, 7) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
        del model
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
