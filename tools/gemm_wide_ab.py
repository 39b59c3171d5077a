"""A/B of family 4's NN launches on eight waves (csrc/gemm8w.hip, policy key 10) against the four-wave kernel of gemm8n.hip (pipelined
schedule, the default) at the step's forward shapes -- plain and with the fused epilogues of the step (bias + QuickGELU with the
derivative copy). Same random bf16 operands, interleaved rounds in one process, HIP events, median and minimum of the rounds.
usage: python tools/gemm_wide_ab.py out.json [rounds]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

M, MKV, W, I = 100864, 117376, 768, 3072
SHAPES = [("vit_fc1_fwd", M, I, W), ("vit_qkv_fwd", M, 3 * W, W), ("kv_proj", MKV, 2 * W, W), ("vit_proj_fwd", M, W, W), ("vit_fc2_fwd", M, W, I),
          ("ast_fc1_fwd", 16512, I, W), ("ast_qkv_fwd", 16512, 3 * W, W), ("ast_proj_fwd", 16512, W, W), ("ast_fc2_fwd", 16512, W, I),
          ("dec_fc1_fwd", 8832, I, W), ("dec_qkv_fwd", 8832, 3 * W, W), ("dec_proj_fwd", 8832, W, W), ("dec_fc2_fwd", 8832, W, I)]


def timeit(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    so = lib.load()
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    res = {"wide_workgroups_per_cu": so.valor_gemm_wide_occupancy(), "narrow_workgroups_per_cu": so.valor_gemm_narrow_occupancy()}
    print(res, flush=True)
    so.valor_gemm_set_policy(8, 1)
    for name, m, n, k in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        A = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
        B = (0.05 * torch.randn((n, k), generator=g)).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(torch.bfloat16).to(dev)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        for tag, fn in (("plain", lambda: K.gemm(A, B, out=out)),
                        ("gelu+deriv", lambda: K.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True))):
            if tag != "plain" and n != I:
                continue
            t = {"n4": [], "n4_m32": [], "w8": []}
            for _ in range(rounds):
                so.valor_gemm_set_policy(10, 0); so.valor_gemm_set_policy(9, 0)
                t["n4"].append(timeit(fn))
                so.valor_gemm_set_policy(9, 1)
                t["n4_m32"].append(timeit(fn))
                so.valor_gemm_set_policy(9, 0); so.valor_gemm_set_policy(10, 1)
                t["w8"].append(timeit(fn))
                so.valor_gemm_set_policy(10, 0)
            fl = 2.0 * m * n * k
            row = {"MNK": [m, n, k]}
            for kk, v in t.items():
                med, mn = sorted(v)[len(v) // 2], min(v)
                row[kk + "_us"] = round(med, 1); row[kk + "_min_us"] = round(mn, 1); row[kk + "_TF"] = round(fl / med / 1e6, 1)
            row["w8_over_n4"] = round(row["n4_us"] / row["w8_us"], 3)
            res[f"{name} {tag}"] = row
            print(name, tag, row, flush=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
