"""A/B of the VideoSwin window-attention backward at the production shapes (VideoSwin-B stages 0..3, per-GPU batch 64, 8 frames -> 4 x side x side
tokens per sample, windows of 196 slots): kernel-family bits of valor_win_attn_set_variant (default 5; + 8 = the first version of the LDS-DMA dQ
pass), HIP events around the whole backward bundle (dQ pass, dK/dV pass, dbias reduce, table gradient), interleaved rounds; the outputs of the two
must be bit-identical. WIN_FWD=1: the forward instead (bit 2 = LDS-DMA forward, + 64 = its first version; e.g. `7 5`: look-ahead DMA forward vs the
register-staged forward -- those two differ in rounding, the check is then a tolerance). usage: python tools/win_bwd_ab.py out.json [variantA variantB]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib, synth  # noqa: E402
from valor_amd.model.valor import VALOR  # noqa: E402

va, vb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5 + 8, 5)      # first version vs default; 21 5: four partitions vs two
dev = torch.device("cuda:0")
so = lib.load()
spec = synth.tiny_swin_spec()
m = VALOR({"dropout": 0.0}, spec=spec, dtype=torch.bfloat16, device=dev)
B, D = 64, int(os.environ.get("WIN_D_TOKENS", "4"))
res = {"variants": [va, vb], "frames_tokens": D}
old = so.valor_win_attn_set_variant(-1)
for stage in range(4):
    heads, side = 4 << stage, 56 >> stage
    C = heads * 32
    for shifted in (False, True):
        geo = m._swin_geometry(D, side, side, shifted)
        g = torch.Generator().manual_seed(stage * 2 + shifted)
        qkv = torch.randn((B * D * side * side, 3 * C), generator=g).to(torch.bfloat16).to(dev)
        table = (0.02 * torch.randn((spec.swin_table, heads), generator=g)).to(torch.bfloat16).to(dev)
        do = torch.randn((qkv.shape[0], C), generator=g).to(torch.bfloat16).to(dev)
        o, lse = K.win_attn_fwd(qkv, geo, table, heads, B)
        FWD = os.environ.get("WIN_FWD", "0") == "1"
        t = {va: [], vb: []}
        outs = {}
        for rnd in range(3):
            for v in (va, vb):
                so.valor_win_attn_set_variant(v)
                run = (lambda: K.win_attn_fwd(qkv, geo, table, heads, B)) if FWD else (lambda: K.win_attn_bwd(qkv, o, lse, do, geo, table, heads, B))
                r0, r1 = run()
                if rnd == 0:
                    outs[v] = (r0.clone(), r1.clone())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run()
                e1.record(); torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 5 * 1e3)
        key = f"stage{stage}_{'shifted' if shifted else 'plain'} ({B * geo['nW']} windows x {heads} heads, N = {geo['N']})"
        res[key] = {f"variant_{va}_us": round(sorted(t[va])[1], 1), f"variant_{vb}_us": round(sorted(t[vb])[1], 1),
                    "ratio": round(sorted(t[va])[1] / sorted(t[vb])[1], 3),
                    "bit_identical": bool(torch.equal(outs[va][0], outs[vb][0]) and torch.equal(outs[va][1], outs[vb][1])),
                    "max_abs_diff": float((outs[va][0].float() - outs[vb][0].float()).abs().max())}
        print(key, res[key], flush=True)
so.valor_win_attn_set_variant(old)
json.dump(res, open(sys.argv[1], "w"), indent=1)
